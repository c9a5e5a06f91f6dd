cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_v4.log 2>&1; rc=$?; tail -4 gpurun_out/pytest_gpu_v4.log
[ $rc -ne 0 ] && exit 1
python bench.py 2>/dev/null | tail -1 > gpurun_out/r02_bench_v4_full.json.log
python bench.py --workload c5_flame_like_1m --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r02_bench_v4_c5_1m.json.log
for f in gpurun_out/r02_bench_v4_*.log; do python -c "
import json,sys
d=json.loads(open('$f').read())
print('$f'.split('v4_')[1], d['value'], d['unit'], d['ms_per_step'], {k:round(v['avg_us']*v.get('launches_per_step',1),1) for k,v in d.get('kernels',{}).items() if 'blend' in k or 'scan' in k})"; done
