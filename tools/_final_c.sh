cd $GRAFT_REPO_ROOT
TAG=r02v4
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof; mkdir -p /tmp/prof $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --profile-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/trace -o $TAG -- $B --steps 30 --warmup 5 > /tmp/prof/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof/pmc_fetch -o $TAG -- $B --steps 4 --warmup 2 > /tmp/prof/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof/pmc_write -o $TAG -- $B --steps 4 --warmup 2 > /tmp/prof/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d /tmp/prof/pmc_sq -o $TAG -- $B --steps 4 --warmup 2 > /tmp/prof/pmc_sq.log 2>&1
python $R/tools/prof_summary.py /tmp/prof $R/gpurun_out/${TAG}_rocprofv3_summary.txt > /dev/null
cd $R
python tools/make_pmc_traffic.py gpurun_out/${TAG}_rocprofv3_summary_traffic.json c2_hotdog_like/trained - profiles/r02_blend_stats_c2.json > /dev/null
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic_v4.json
python bench.py 2>/dev/null | tail -1 > gpurun_out/r02_bench_v4_full.json.log
python bench.py --loss l1_ssim --optimizer fused_adam --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r02_bench_v4_full_iteration.json.log
python bench.py --workload c5_flame_like_1m --mode animate --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r02_bench_v4_c5_1m_animate.json.log
python bench.py --workload c4_ficus_like --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r02_bench_v4_c4_single_gpu.json.log
grep -E "^(blend|tile_|preprocess|emit|mesh)" gpurun_out/${TAG}_rocprofv3_summary.txt | head -16 | cut -c1-130
for f in gpurun_out/r02_bench_v4_*.log; do python -c "
import json,sys
d=json.loads(open('$f').read())
print('$f'.split('v4_')[1], d['value'], d['unit'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'))"; done
