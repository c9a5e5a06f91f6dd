# round 5, GPU call 14: unit-level opacity exponent in the fixed-point table (a field's scale is one constant per lane and unit)
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=r05n
bash tools/ab.sh $T "-" "-" "-"
timeout 1500 python -m pytest tests/test_gpu_raster.py tests/test_gpu_deterministic.py tests/test_gpu_negative_controls.py tests/test_gpu_c4.py tests/test_gpu_training.py tests/test_gpu_fixed_point.py -m gpu -q --maxfail=6 2>&1 | tail -8 > gpurun_out/${T}_pytest.log; tail -3 gpurun_out/${T}_pytest.log | cut -c1-300
timeout 500 python tools/fuzz_parity.py 150 65000 > gpurun_out/${T}_fuzz_150cases.log 2>&1; tail -2 gpurun_out/${T}_fuzz_150cases.log | cut -c1-300
timeout 300 python tools/fuzz_parity.py 60 66000 det > gpurun_out/${T}_fuzz_60cases_det_strict.log 2>&1; tail -2 gpurun_out/${T}_fuzz_60cases_det_strict.log | cut -c1-300
