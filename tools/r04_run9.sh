cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=r04i
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline"
for v in "GMS_INLINE_SCAN=0" "GMS_INLINE_SCAN=1"; do
  tag=$(echo "$v" | tr ' =' '__')
  env $v $B > gpurun_out/${T}_ab_${tag}.json.log 2> gpurun_out/${T}_ab_${tag}.err
  echo "$v: $(grep -o '"value": [0-9.]*' gpurun_out/${T}_ab_${tag}.json.log) $(grep -o '"emit_instances": {"avg_us": [0-9.]*' gpurun_out/${T}_ab_${tag}.json.log) $(grep -o '"tile_scan": {"avg_us": [0-9.]*' gpurun_out/${T}_ab_${tag}.json.log) $(grep -o '"binning": {[^}]*}' gpurun_out/${T}_ab_${tag}.json.log | cut -c1-160)"
  env $v $B --workload c5_flame_like_1m --steps 40 > gpurun_out/${T}_c5_${tag}.log 2>&1; echo "c5 $v: $(grep -o '"value": [0-9.]*' gpurun_out/${T}_c5_${tag}.log) $(grep -o '"emit_instances": {"avg_us": [0-9.]*' gpurun_out/${T}_c5_${tag}.log)"
  env $v $B --workload c5_flame_like_1m --steps 60 --mode animate > gpurun_out/${T}_c5a_${tag}.log 2>&1; echo "c5 animate $v: $(grep -o '"value": [0-9.]*' gpurun_out/${T}_c5a_${tag}.log)"
done
python -m pytest tests/test_gpu_raster.py tests/test_gpu_deterministic.py tests/test_gpu_training.py -q -x 2>&1 | tail -6 > gpurun_out/${T}_pytest.log; tail -3 gpurun_out/${T}_pytest.log
