#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace + separate PMC passes of bench.py,
# condensed into gpurun_out/<tag>_*.txt / _traffic.json (copy the ones to keep into profiles/).
# usage: tools/collect_profiles.sh <tag>
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf /tmp/prof; mkdir -p /tmp/prof $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --profile-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/trace -o $TAG -- $B --steps 30 --warmup 5 > /tmp/prof/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof/pmc_fetch -o $TAG -- $B --steps 4 --warmup 2 > /tmp/prof/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof/pmc_write -o $TAG -- $B --steps 4 --warmup 2 > /tmp/prof/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d /tmp/prof/pmc_sq -o $TAG -- $B --steps 4 --warmup 2 > /tmp/prof/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d /tmp/prof/pmc_sq2 -o $TAG -- $B --steps 4 --warmup 2 > /tmp/prof/pmc_sq2.log 2>&1
# ROCTX ranges of the C-ABI entry points (GMS_ROCTX=1) next to the kernels: marker + kernel trace, no counters
GMS_ROCTX=1 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d /tmp/prof/marker -o $TAG -- $B --steps 10 --warmup 2 > /tmp/prof/marker.log 2>&1
python $R/tools/prof_summary.py /tmp/prof $R/gpurun_out/${TAG}_rocprofv3_summary.txt > /dev/null
(echo "# rocprofv3 --kernel-trace --marker-trace --stats with GMS_ROCTX=1 (ranges = C-ABI entry points)"; for f in $(find /tmp/prof/marker -name "*marker*stats*.csv" -o -name "*marker_api_stats.csv" | head -3); do echo "## $(basename $f)"; head -12 $f; done) > $R/gpurun_out/${TAG}_roctx_ranges.txt 2>/dev/null
grep -E "^(blend|tile_s|preprocess|emit|mesh|fill_units)" $R/gpurun_out/${TAG}_rocprofv3_summary.txt | head -16 | cut -c1-140
cat $R/gpurun_out/${TAG}_rocprofv3_summary_traffic.json | head -80
