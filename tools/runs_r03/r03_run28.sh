cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { python bench.py --steps 60 --warmup 10 --no-cpu-baseline --workload c5_flame_like_1m --mode animate 2>&1 | grep -o '"value": [0-9.]*, "unit": "[a-z/]*"'; }
echo default; run
for L in 256 1024 4096 32768; do echo "MICRO=0 L=$L"; GMS_MICRO=0 GMS_SEG_LEN=$L run; done
echo "MICRO=0 L=32768 DEEP=0"; GMS_MICRO=0 GMS_SEG_LEN=32768 GMS_DEEP=0 run
