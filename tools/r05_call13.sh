# round 5, GPU call 13: the micro-tile kernels on the deep config-5 scene (they changed this round), and the headline line three times
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=r05m
for E in "GMS_MICRO=1" "X=1"; do
  echo "== $E c5_500k fwd+bwd" | tee -a gpurun_out/${T}_c5.txt
  env $E python bench.py --steps 60 --warmup 10 --no-cpu-baseline --workload c5_flame_like_500k --profile-steps 0 2>/dev/null | tail -1 | cut -c1-200 | tee -a gpurun_out/${T}_c5.txt
  echo "== $E c5_500k animate" | tee -a gpurun_out/${T}_c5.txt
  env $E python bench.py --steps 100 --warmup 20 --no-cpu-baseline --workload c5_flame_like_500k --mode animate 2>/dev/null | tail -1 | cut -c1-200 | tee -a gpurun_out/${T}_c5.txt
done
for k in 1 2 3; do python bench.py --steps 200 --warmup 20 --no-cpu-baseline --profile-steps 0 2>/dev/null | tail -1 | cut -c1-160 | tee -a gpurun_out/${T}_headline_x3.txt; done
