set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export GMS_MICRO=1
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_negative_controls.py -q -x 2>&1 | tail -40 > gpurun_out/r03_pytest_micro_a.log
tail -5 gpurun_out/r03_pytest_micro_a.log
for L in 512 256 1024; do
  GMS_SEG_LEN=$L timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r03_bench_micro_L$L.log 2> gpurun_out/r03_bench_micro_L$L.err
done
GMS_TRIP=4 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r03_bench_micro_trip4.log 2>&1
GMS_TRIP=1 GMS_TRIP_BWD=1 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r03_bench_micro_trip1.log 2>&1
timeout 600 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_c4.py -q 2>&1 | tail -15 > gpurun_out/r03_pytest_micro_b.log
tail -3 gpurun_out/r03_pytest_micro_b.log
grep -h -o '"value": [0-9.]*' gpurun_out/r03_bench_micro_*.log
