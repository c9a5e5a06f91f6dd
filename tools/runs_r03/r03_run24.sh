cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 0 1; do for tb in 2 4; do echo "PRIV=$v TRIP_BWD=$tb"; GMS_BWD_PRIV=$v GMS_TRIP_BWD=$tb python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*, "unit": "iters/s"\|"avg_launch_us": [0-9.]*' | head -2 | tr '\n' ' '; echo; done; done
echo "PRIV=1 TRIP_BWD=1"; GMS_BWD_PRIV=1 GMS_TRIP_BWD=1 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*, "unit": "iters/s"\|"avg_launch_us": [0-9.]*' | head -2 | tr '\n' ' '; echo
timeout 600 python -m pytest tests/test_gpu_raster.py -q -x -k "knob" 2>&1 | tail -2
