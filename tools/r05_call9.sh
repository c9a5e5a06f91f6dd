# round 5, GPU call 9: interleaved row reductions (2 and 4 entries per trip), bench --gpus 8 on the shared-GPU gloo path, full GPU suite
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=r05i
bash tools/ab.sh $T "-" "GMS_TRIP_BWD=4" "-" "GMS_TRIP_BWD=4" "GMS_BWD_FIXED=0"
timeout 900 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 > gpurun_out/${T}_bench_gpus8_shared.json.log 2>&1; tail -1 gpurun_out/${T}_bench_gpus8_shared.json.log | cut -c1-600
timeout 2400 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -30 > gpurun_out/${T}_pytest_gpu.log; tail -6 gpurun_out/${T}_pytest_gpu.log | cut -c1-300
