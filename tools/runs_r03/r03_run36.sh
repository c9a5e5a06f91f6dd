cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 120 python -m pytest tests/test_gpu_raster.py -q -k "very_deep" > gpurun_out/r03_deep_test_fixed.log 2>&1; tail -3 gpurun_out/r03_deep_test_fixed.log
grep -n "Error\|error\|assert" gpurun_out/r03_deep_test_fixed.log | head -20 | cut -c1-400
