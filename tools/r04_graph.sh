# hipGraph animation: tests + renders/s eager vs graph replay.  bash tools/r04_graph.sh
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_graph.py -q -x 2>&1 | tail -30 > gpurun_out/r04o_pytest.log; tail -30 gpurun_out/r04o_pytest.log | cut -c1-300
for w in small; do
  for g in "" "--graph"; do
    timeout 120 python bench.py --no-cpu-baseline --steps 100 --warmup 20 --workload $w --mode animate $g > gpurun_out/r04o_anim_${w}${g}.log 2>&1
    echo "$w $g: $(grep -o '"value": [0-9.]*' gpurun_out/r04o_anim_${w}${g}.log) $(grep -o '"graph": {[^}]*}' gpurun_out/r04o_anim_${w}${g}.log | cut -c1-200)"
  done
done
