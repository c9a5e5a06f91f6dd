cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=r04m
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline"
for v in "GMS_BWD_STAGE=0" "GMS_BWD_STAGE=1"; do
  tag=$(echo "$v" | tr ' =' '__')
  env $v $B > gpurun_out/${T}_ab_${tag}.json.log 2> gpurun_out/${T}_ab_${tag}.err
  echo "$v: $(grep -o '"value": [0-9.]*' gpurun_out/${T}_ab_${tag}.json.log) $(grep -o '"blend_bwd": {"avg_us": [0-9.]*' gpurun_out/${T}_ab_${tag}.json.log)"
done
GMS_BWD_STAGE=1 python -m pytest tests/test_gpu_raster.py tests/test_gpu_negative_controls.py tests/test_gpu_training.py tests/test_gpu_mesh.py -q -x 2>&1 | tail -6 > gpurun_out/${T}_pytest.log; tail -3 gpurun_out/${T}_pytest.log
