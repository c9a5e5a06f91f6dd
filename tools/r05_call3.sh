# round 5, GPU call 3: the suite on the cleaned tree (resident units only, LDS-DMA preprocess, fused mesh frames with contraction-free
# operators, rank merge, two-leg gradient gate), integer LDS atomics in the microbenchmark, A/B of the merge
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=r05c
timeout 120 tools/lds_bench.bin > gpurun_out/${T}_lds_bench.txt 2>&1; tail -4 gpurun_out/${T}_lds_bench.txt | cut -c1-200
timeout 2400 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -40 > gpurun_out/${T}_pytest_gpu.log; tail -12 gpurun_out/${T}_pytest_gpu.log | cut -c1-300
bash tools/ab.sh $T "-" "GMS_MERGE=path" "-" "GMS_MERGE=path"
for E in "GMS_ANIMATE_FUSED=1" "GMS_ANIMATE_FUSED=0"; do
  echo "== $E" | tee -a gpurun_out/${T}_animate.txt
  env $E python bench.py --steps 100 --warmup 20 --no-cpu-baseline --workload c5_flame_like_500k --mode animate 2>/dev/null | tail -1 | cut -c1-300 | tee -a gpurun_out/${T}_animate.txt
done
timeout 400 python tools/fuzz_parity.py 100 53000 > gpurun_out/${T}_fuzz_100cases.log 2>&1; tail -3 gpurun_out/${T}_fuzz_100cases.log | cut -c1-300
timeout 300 python tools/fuzz_parity.py 40 54000 det > gpurun_out/${T}_fuzz_40cases_det_strict.log 2>&1; tail -3 gpurun_out/${T}_fuzz_40cases_det_strict.log | cut -c1-300
