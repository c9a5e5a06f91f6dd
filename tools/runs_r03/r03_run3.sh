set -x
R=$GRAFT_REPO_ROOT
cd $R
export TMPDIR=/tmp
export GMS_MICRO=1
for L in 128 192; do
  GMS_SEG_LEN=$L timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r03_bench_micro_L$L.log 2> gpurun_out/r03_bench_micro_L$L.err
done
export GMS_SEG_LEN=256
rm -rf /tmp/prof; mkdir -p /tmp/prof
cd /tmp
B="python $R/bench.py --no-cpu-baseline --profile-steps 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/trace -o m256 -- $B --steps 30 --warmup 5 > /tmp/prof/trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d /tmp/prof/pmc_sq -o m256 -- $B --steps 4 --warmup 2 > /tmp/prof/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d /tmp/prof/pmc_sq2 -o m256 -- $B --steps 4 --warmup 2 > /tmp/prof/pmc_sq2.log 2>&1
python $R/tools/prof_summary.py /tmp/prof $R/gpurun_out/r03_micro256_rocprofv3_summary.txt > /dev/null
cd $R
unset GMS_SEG_LEN
for M in 1 0; do
  GMS_MICRO=$M timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --workload c5_flame_like_1m --profile-steps 10 > gpurun_out/r03_c5_micro$M.log 2>&1
  GMS_MICRO=$M timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --workload c5_flame_like_1m --mode animate > gpurun_out/r03_c5_anim_micro$M.log 2>&1
done
GMS_MICRO=1 GMS_SEG_LEN=1024 timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --workload c5_flame_like_1m --profile-steps 10 > gpurun_out/r03_c5_micro1_L1024.log 2>&1
grep -h -o '"value": [0-9.]*' gpurun_out/r03_bench_micro_L128.log gpurun_out/r03_bench_micro_L192.log gpurun_out/r03_c5_*.log
grep -E "micro|blend" gpurun_out/r03_micro256_rocprofv3_summary.txt | cut -c1-250 | head -30
