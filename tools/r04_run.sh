cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r04a_pytest_gpu.log
tail -3 gpurun_out/r04a_pytest_gpu.log
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r04a_bench.json.log 2> gpurun_out/r04a_bench.err
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --workload c5_flame_like_1m --profile-steps 10 > gpurun_out/r04a_bench_c5.json.log 2>&1
grep -h -o '"value": [0-9.]*' gpurun_out/r04a_bench*.json.log
