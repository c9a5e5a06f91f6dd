cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_bench.py tests/test_gpu_deterministic.py tests/test_gpu_mesh.py -q -x 2>&1 | tail -8 > gpurun_out/r04g_pytest.log; tail -3 gpurun_out/r04g_pytest.log
bash tools/r04_prof.sh r04_det "GAMES_HIP_DETERMINISTIC=1" | head -12
GAMES_HIP_DETERMINISTIC=1 python tools/train_c3.py > gpurun_out/r04_train_c3_7000iters_deterministic.json.log 2> gpurun_out/r04_train_c3_det.err; cut -c1-500 gpurun_out/r04_train_c3_7000iters_deterministic.json.log
