cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_training.py -q -x 2>&1 | tail -15 > gpurun_out/r04e_pytest_training.log; tail -3 gpurun_out/r04e_pytest_training.log
python tools/train_c3.py > gpurun_out/r04_train_c3_7000iters.json.log 2> gpurun_out/r04_train_c3.err; cut -c1-600 gpurun_out/r04_train_c3_7000iters.json.log
GAMES_HIP_DETERMINISTIC=1 python tools/train_c3.py > gpurun_out/r04_train_c3_7000iters_deterministic.json.log 2> gpurun_out/r04_train_c3_det.err; cut -c1-400 gpurun_out/r04_train_c3_7000iters_deterministic.json.log
