# kernel-trace profile of bench.py under an environment: bash tools/prof_bench.sh TAG "ENV=..." [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; ENVS=$2; shift 2
rm -rf /tmp/prof_$TAG; mkdir -p /tmp/prof_$TAG $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
env $ENVS rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/trace -o $TAG -- python $R/bench.py --no-cpu-baseline --profile-steps 0 --steps 30 --warmup 5 "$@" > /tmp/prof_$TAG/trace.log 2>&1
python $R/tools/prof_summary.py /tmp/prof_$TAG $R/gpurun_out/${TAG}_rocprofv3_summary.txt > /dev/null
head -30 $R/gpurun_out/${TAG}_rocprofv3_summary.txt | cut -c1-150
