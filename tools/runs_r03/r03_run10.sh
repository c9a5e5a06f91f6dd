set -x
R=$GRAFT_REPO_ROOT
cd $R
export TMPDIR=/tmp
export GMS_MICRO=1
export GMS_SEG_LEN=256
rm -rf /tmp/prof; mkdir -p /tmp/prof
cd /tmp
B="python $R/bench.py --no-cpu-baseline --profile-steps 0"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d /tmp/prof/pmc_sq -o m3 -- $B --steps 4 --warmup 2 > /tmp/prof/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d /tmp/prof/pmc_sq2 -o m3 -- $B --steps 4 --warmup 2 > /tmp/prof/pmc_sq2.log 2>&1
python $R/tools/prof_summary.py /tmp/prof $R/gpurun_out/r03_micro3_rocprofv3_summary.txt > /dev/null
grep -E "micro|blend" $R/gpurun_out/r03_micro3_rocprofv3_summary.txt | sed 's/  */ /g' | cut -c1-600
tail -3 /tmp/prof/pmc_sq2.log
