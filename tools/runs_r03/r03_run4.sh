set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export GMS_MICRO=1
export GMS_SEG_LEN=256
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_negative_controls.py -q -x 2>&1 | tail -30 > gpurun_out/r03_pytest_micro2_a.log
tail -4 gpurun_out/r03_pytest_micro2_a.log
for L in 256 512 128; do
  GMS_SEG_LEN=$L timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r03_bench_micro2_L$L.log 2> gpurun_out/r03_bench_micro2_L$L.err
done
GMS_TRIP_BWD=4 GMS_TRIP=4 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r03_bench_micro2_trip4.log 2>&1
GMS_TRIP_BWD=1 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r03_bench_micro2_bwdtrip1.log 2>&1
grep -h -o '"value": [0-9.]*' gpurun_out/r03_bench_micro2_*.log
