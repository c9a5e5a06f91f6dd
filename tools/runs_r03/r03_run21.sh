cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
for s in 5113 5179 5312 5337 5364 5526 5838; do
  echo "== seed $s"; timeout 200 python tools/fuzz_parity.py 1 $s 2>&1 | grep -E "^ok|MISMATCH|worst|cond|COND" | cut -c1-400
done > gpurun_out/r03_fuzz_bad7_newrule.txt 2>&1
cat gpurun_out/r03_fuzz_bad7_newrule.txt
timeout 900 python -m pytest tests/test_gpu_negative_controls.py tests/test_gpu_raster.py -q -x 2>&1 | tail -5
timeout 600 python tools/fuzz_parity.py 600 7000 > gpurun_out/r03_fuzz_600_final.log 2>&1
tail -4 gpurun_out/r03_fuzz_600_final.log | cut -c1-400; grep -c MISMATCH gpurun_out/r03_fuzz_600_final.log; grep MISMATCH gpurun_out/r03_fuzz_600_final.log | cut -c1-600 | head
