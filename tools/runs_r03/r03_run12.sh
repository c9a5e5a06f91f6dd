set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r03_pytest_c.log
tail -4 gpurun_out/r03_pytest_c.log
python bench.py --steps 200 --warmup 20 > gpurun_out/r03_bench_c.log 2> gpurun_out/r03_bench_c.err
timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --workload c5_flame_like_1m --profile-steps 10 > gpurun_out/r03_c5c.log 2>&1
bash tools/collect_profiles.sh r03c > gpurun_out/r03_collect_c.log 2>&1
grep -h -o '"value": [0-9.]*' gpurun_out/r03_bench_c.log gpurun_out/r03_c5c.log
grep -E "^(micro|blend|tile_s|preprocess|emit|mesh)" gpurun_out/r03c_rocprofv3_summary.txt | head -16 | cut -c1-130
