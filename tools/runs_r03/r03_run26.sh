cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*, "unit": "iters/s"\|"avg_launch_us": [0-9.]*' | head -2 | tr '\n' ' '; echo; }
for qb in 16 8; do for tb in 2 4 1; do echo "QB=$qb TRIP_BWD=$tb"; GMS_BWD_QB=$qb GMS_TRIP_BWD=$tb run; done; done
echo "QB=8 again"; GMS_BWD_QB=8 run; echo "QB=16 again"; GMS_BWD_QB=16 run
timeout 600 python -m pytest tests/test_gpu_raster.py -q -x -k "knob" 2>&1 | tail -2
