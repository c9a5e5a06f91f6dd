cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for w in 2 3; do GMS_PRE_WAVES=$w python -m pytest tests/test_gpu_raster.py -q -x -k "parity or split_sh or precomputed or full_size" 2>&1 | tail -2; done
AB="GMS_PRE_WAVES=1;GMS_PRE_WAVES=2;GMS_PRE_WAVES=3" T=r04h
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline"
IFS=';' read -ra VARS <<< "$AB"
for v in "${VARS[@]}"; do
  tag=$(echo "$v" | tr ' =' '__')
  env $v $B > gpurun_out/${T}_ab_${tag}.json.log 2> gpurun_out/${T}_ab_${tag}.err
  echo "$v: $(grep -o '"value": [0-9.]*' gpurun_out/${T}_ab_${tag}.json.log) $(grep -o '"preprocess_fwd": {"avg_us": [0-9.]*' gpurun_out/${T}_ab_${tag}.json.log) $(grep -o '"mesh_fwd": {"avg_us": [0-9.]*' gpurun_out/${T}_ab_${tag}.json.log)"
done
for v in "${VARS[@]}"; do env $v $B --workload c5_flame_like_1m --steps 40 > gpurun_out/${T}_c5.log 2>&1; echo "c5 $v: $(grep -o '"value": [0-9.]*' gpurun_out/${T}_c5.log) $(grep -o '"preprocess_fwd": {"avg_us": [0-9.]*' gpurun_out/${T}_c5.log)"; done
