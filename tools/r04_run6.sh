cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_bench.py -q -x 2>&1 | tail -15 > gpurun_out/r04f_pytest_bench.log; tail -4 gpurun_out/r04f_pytest_bench.log
bash tools/r04_prof.sh r04_det "GAMES_HIP_DETERMINISTIC=1"
GAMES_HIP_DETERMINISTIC=1 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --profile-steps 0 > gpurun_out/r04f_bench_det.json.log 2>&1; grep -o '"value": [0-9.]*' gpurun_out/r04f_bench_det.json.log
GAMES_HIP_DETERMINISTIC=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 --workload c5_flame_like_1m > gpurun_out/r04f_bench_det_c5.json.log 2>&1; grep -o '"value": [0-9.]*' gpurun_out/r04f_bench_det_c5.json.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r04f_bench_default.json.log 2> gpurun_out/r04f_bench_default.err; cut -c1-300 gpurun_out/r04f_bench_default.json.log; grep -o '"stages": {.*"whole_iteration"' gpurun_out/r04f_bench_default.json.log | cut -c1-1200; grep -o '"dense_torch_raster_c1": {[^}]*}' gpurun_out/r04f_bench_default.json.log
