# Round-4 survey run (GPU box, no source change): per-launch profile of config 5 (S = 50), env-only A/B at that size, more fuzz seeds.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=r04s
B5="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --profile-steps 0 --workload c5_flame_like_500k"
rm -rf /tmp/prof5; mkdir -p /tmp/prof5
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof5/trace -o $T -- $B5 --steps 30 --warmup 5 > /tmp/prof5/trace.log 2>&1)
python tools/prof_summary.py /tmp/prof5 gpurun_out/${T}_c5_500k_rocprofv3_summary.txt > /dev/null
python tools/step_sequence.py /tmp/prof5/trace gpurun_out/${T}_c5_500k_step_sequence.txt | cut -c1-110
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof5a/trace -o $T -- $B5 --mode animate --steps 30 --warmup 5 > /tmp/prof5/trace_a.log 2>&1)
python tools/step_sequence.py /tmp/prof5a/trace gpurun_out/${T}_c5_500k_animate_step_sequence.txt | cut -c1-110
for E in "X=0" "GMS_MICRO=1" "GMS_SEG_LEN=256" "GMS_SEG_LEN=1024" "GMS_INLINE_SCAN=0"; do
  echo "== $E"; env $E timeout 200 $B5 --steps 40 --warmup 8 2>/dev/null | grep -o '"value": [0-9.]*'
  env $E timeout 200 $B5 --mode animate --steps 40 --warmup 8 2>/dev/null | grep -o '"value": [0-9.]*'
done 2>&1 | tee gpurun_out/${T}_c5_500k_env_ab.txt
timeout 400 python tools/fuzz_parity.py 300 31000 > gpurun_out/${T}_fuzz_300cases_seed31000.log 2>&1; tail -4 gpurun_out/${T}_fuzz_300cases_seed31000.log | cut -c1-250
