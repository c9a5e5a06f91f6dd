cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1000 python tools/fuzz_parity.py 800 9000 > gpurun_out/r03_fuzz_800_final.log 2>&1
tail -4 gpurun_out/r03_fuzz_800_final.log | cut -c1-400; grep -c MISMATCH gpurun_out/r03_fuzz_800_final.log; grep MISMATCH gpurun_out/r03_fuzz_800_final.log | cut -c1-700 | head -5
