cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for i in 1 2; do python bench.py --steps 60 --warmup 10 --no-cpu-baseline --workload c5_flame_like_1m --mode animate 2>&1 | grep -o '"value": [0-9.]*, "unit": "[a-z/]*"'; done
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --workload c5_flame_like_1m --mode animate > gpurun_out/r03v3_bench_c5_1m_animate.json.log 2>&1; grep -o '"value": [0-9.]*, "unit": "[a-z/]*"' gpurun_out/r03v3_bench_c5_1m_animate.json.log
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --profile-steps 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "[a-z/]*"'
timeout 600 python -m pytest tests/test_gpu_raster.py -q -x -k "capacity or knob or hint" 2>&1 | tail -2
