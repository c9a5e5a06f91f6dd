# round 5, GPU call 2: LDS microbenchmark; full GPU suite on the round's changes (resident units, LDS-DMA preprocess, fused mesh
# frames, quirk switch, calibrated kernel table); A/B of the preprocess staging and of the fused animated frame
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=r05b
timeout 120 tools/lds_bench.bin > gpurun_out/${T}_lds_bench.txt 2>&1; cat gpurun_out/${T}_lds_bench.txt | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/${T}_pytest_gpu.log; tail -6 gpurun_out/${T}_pytest_gpu.log
bash tools/ab.sh $T "-" "GMS_PRE_DMA=0" "-" "GMS_PRE_DMA=0"
for E in "GMS_ANIMATE_FUSED=1" "GMS_ANIMATE_FUSED=0" "GMS_ANIMATE_FUSED=1 GMS_PRE_DMA=0"; do
  echo "== $E" | tee -a gpurun_out/${T}_animate.txt
  env $E python bench.py --steps 100 --warmup 20 --no-cpu-baseline --workload c5_flame_like_500k --mode animate 2>/dev/null | tail -1 | cut -c1-420 | tee -a gpurun_out/${T}_animate.txt
done
timeout 300 python tools/fuzz_parity.py 60 52000 > gpurun_out/${T}_fuzz_60cases.log 2>&1; tail -3 gpurun_out/${T}_fuzz_60cases.log | cut -c1-300
