cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 25 python -m pytest tests/test_gpu_raster.py -q -x -k "huge_thin" 2>&1 | tail -1 | cut -c1-300
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --workload c5_flame_like_1m --profile-steps 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "[a-z/]*"'
T=r03v8
bash tools/collect_profiles.sh $T > gpurun_out/${T}_collect.log 2>&1
python tools/make_pmc_traffic.py gpurun_out/${T}_rocprofv3_summary_traffic.json c2_hotdog_like/trained - profiles/r02_blend_stats_c2.json > gpurun_out/${T}_make_pmc.log 2>&1
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
echo done
