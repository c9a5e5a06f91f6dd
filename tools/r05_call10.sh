# round 5, GPU call 10: unconditional integer adds in micro_bwd
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=r05j
bash tools/ab.sh $T "-" "-" "-"
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_deterministic.py tests/test_gpu_negative_controls.py -m gpu -q --maxfail=6 2>&1 | tail -8 > gpurun_out/${T}_pytest.log; tail -3 gpurun_out/${T}_pytest.log | cut -c1-300
timeout 400 python tools/fuzz_parity.py 100 60000 > gpurun_out/${T}_fuzz_100cases.log 2>&1; tail -2 gpurun_out/${T}_fuzz_100cases.log | cut -c1-300
