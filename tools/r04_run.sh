# Round-4 GPU run (bash tools/r04_run.sh TAG [what]): what = tests | ab | all ; AB="envs;envs;..." overrides the variant list
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=${1:-r04}; WHAT=${2:-all}
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline"
AB=${AB:-"GMS_BWD_PIPE=0;GMS_BWD_PIPE=1;GMS_BWD_PIPE=2;GMS_BWD_PIPE=1 GMS_TRIP_BWD=1"}
if [ "$WHAT" = "all" ] || [ "$WHAT" = "ab" ]; then
  IFS=';' read -ra VARS <<< "$AB"
  for v in "${VARS[@]}"; do
    tag=$(echo "$v" | tr ' =' '__')
    env $v $B > gpurun_out/${T}_ab_${tag}.json.log 2> gpurun_out/${T}_ab_${tag}.err
    echo "$v: $(grep -o '"value": [0-9.]*' gpurun_out/${T}_ab_${tag}.json.log) $(grep -o '"blend_bwd": {"avg_us": [0-9.]*' gpurun_out/${T}_ab_${tag}.json.log) $(grep -o '"blend_head": {"avg_us": [0-9.]*' gpurun_out/${T}_ab_${tag}.json.log) $(grep -o '"blend_fwd": {"avg_us": [0-9.]*' gpurun_out/${T}_ab_${tag}.json.log)"
  done
fi
if [ "$WHAT" = "all" ] || [ "$WHAT" = "tests" ]; then
  rm -f gpurun_out/parity_report.jsonl
  python -m pytest tests -m gpu -q ${PYTEST_ARGS:--x} 2>&1 | tail -40 > gpurun_out/${T}_pytest_gpu.log
  tail -6 gpurun_out/${T}_pytest_gpu.log
fi
