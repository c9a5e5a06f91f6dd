cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --workload c5_flame_like_1m --mode animate 2>&1 | grep -o '"value": [0-9.]*, "unit": "[a-z/]*"'
GMS_TRIP=2 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --workload c5_flame_like_1m --mode animate 2>&1 | grep -o '"value": [0-9.]*, "unit": "[a-z/]*"'
python bench.py --steps 200 --warmup 30 --no-cpu-baseline --workload c5_flame_like_1m --mode animate 2>&1 | grep -o '"value": [0-9.]*, "unit": "[a-z/]*"'
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -o an -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --workload c5_flame_like_1m --mode animate > /tmp/pa.log 2>&1
f=$(find /tmp/pa -name "*kernel_stats.csv" | head -1); head -16 $f | cut -c1-160
