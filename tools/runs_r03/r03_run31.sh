cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { python bench.py --steps 80 --warmup 10 --no-cpu-baseline --workload c5_flame_like_1m --mode animate 2>&1 | grep -o '"value": [0-9.]*, "unit": "[a-z/]*"'; }
for L in 256 512 768 1024 1536 2048; do echo "MICRO=0 L=$L"; GMS_MICRO=0 GMS_SEG_LEN=$L run; done
echo "fwd+bwd L=512"; GMS_MICRO=0 GMS_SEG_LEN=512 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --workload c5_flame_like_1m --profile-steps 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "[a-z/]*"'
echo "fwd+bwd L=1024"; GMS_MICRO=0 GMS_SEG_LEN=1024 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --workload c5_flame_like_1m --profile-steps 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "[a-z/]*"'
