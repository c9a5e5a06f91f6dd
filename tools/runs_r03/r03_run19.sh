cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python tools/fuzz_parity.py 1000 5000 > gpurun_out/r03_fuzz_1000.log 2>&1
tail -3 gpurun_out/r03_fuzz_1000.log
grep -c "^ok" gpurun_out/r03_fuzz_1000.log; grep -E "MISMATCH|ERROR" gpurun_out/r03_fuzz_1000.log | head -5
