# round 5, GPU call 8: fixed-point table (float scaling + double magic number, bounds from the walk's own loads), deterministic mode on it
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=r05h
bash tools/ab.sh $T "-" "GMS_BWD_FIXED=0" "-" "GMS_BWD_FIXED=0" "GAMES_HIP_DETERMINISTIC=1"
X=$GRAFT_REPO_ROOT/gaussian-mesh-splatting_amd/lib_exp
LD_LIBRARY_PATH=$X:$LD_LIBRARY_PATH GMSPLAT_LIB=$X/libgmsplat.so GMS_DBG=1024 timeout 300 python tools/micro_phases.py > gpurun_out/${T}_phases_fixed.txt 2>&1; tail -14 gpurun_out/${T}_phases_fixed.txt | cut -c1-300
timeout 1500 python -m pytest tests/test_gpu_raster.py tests/test_gpu_deterministic.py tests/test_gpu_negative_controls.py tests/test_gpu_c4.py tests/test_gpu_training.py -m gpu -q --maxfail=6 2>&1 | tail -30 > gpurun_out/${T}_pytest.log; tail -8 gpurun_out/${T}_pytest.log | cut -c1-400
timeout 500 python tools/fuzz_parity.py 150 58000 > gpurun_out/${T}_fuzz_150cases.log 2>&1; tail -4 gpurun_out/${T}_fuzz_150cases.log | cut -c1-300
timeout 300 python tools/fuzz_parity.py 60 59000 det > gpurun_out/${T}_fuzz_60cases_det_strict.log 2>&1; tail -3 gpurun_out/${T}_fuzz_60cases_det_strict.log | cut -c1-300
