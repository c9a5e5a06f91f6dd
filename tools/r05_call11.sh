# round 5, GPU call 11: preprocess_bwd with the linear LDS-DMA staging of the SH rows
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=r05k
bash tools/ab.sh $T "-" "GMS_PRE_BWD_LINEAR=0" "-" "GMS_PRE_BWD_LINEAR=0"
timeout 1500 python -m pytest tests/test_gpu_raster.py tests/test_gpu_shfactor.py tests/test_gpu_training.py tests/test_gpu_c4.py tests/test_gpu_deterministic.py -m gpu -q --maxfail=6 2>&1 | tail -8 > gpurun_out/${T}_pytest.log; tail -3 gpurun_out/${T}_pytest.log | cut -c1-300
timeout 300 python tools/fuzz_parity.py 80 64000 > gpurun_out/${T}_fuzz_80cases.log 2>&1; tail -2 gpurun_out/${T}_fuzz_80cases.log | cut -c1-300
