cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { python bench.py --steps 150 --warmup 20 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*, "unit": "iters/s"\|"avg_launch_us": [0-9.]*' | head -2 | tr '\n' ' '; echo; }
for pad in 0 6000 14000 27000; do echo "atomic pad=$pad"; GMS_BWD_PADLDS=$pad run; done
for pad in 0 26000; do echo "PRIV pad=$pad"; GMS_BWD_PRIV=1 GMS_BWD_PADLDS=$pad run; done
