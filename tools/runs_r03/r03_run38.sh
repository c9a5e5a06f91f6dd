cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 70 python -m pytest tests/test_gpu_raster.py -q -x -k "huge_thin or very_deep or forward_backward_parity" > gpurun_out/r03_cull_fix_tests.log 2>&1; tail -3 gpurun_out/r03_cull_fix_tests.log | cut -c1-600
T=r03v7
bash tools/collect_profiles.sh $T > gpurun_out/${T}_collect.log 2>&1
python tools/make_pmc_traffic.py gpurun_out/${T}_rocprofv3_summary_traffic.json c2_hotdog_like/trained - profiles/r02_blend_stats_c2.json > gpurun_out/${T}_make_pmc.log 2>&1
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['roofline']['traffic'], d['roofline']['pmc'][:100])"
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --workload c5_flame_like_1m --profile-steps 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "[a-z/]*"'
