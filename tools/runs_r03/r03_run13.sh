set -x
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof; mkdir -p /tmp/prof
GMS_SH_STREAM=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof/trace -o sh -- python $R/bench.py --no-cpu-baseline --profile-steps 0 --steps 20 --warmup 5 > /tmp/prof/trace.log 2>&1
f=$(find /tmp/prof/trace -name "*kernel_trace.csv" | head -1)
python - "$f" > $R/gpurun_out/r03_trace_shstream.txt <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last 40 kernels
base=int(rows[-60]["Start_Timestamp"])
for r in rows[-60:]:
    print(f'{(int(r["Start_Timestamp"])-base)/1000:9.1f} {(int(r["End_Timestamp"])-base)/1000:9.1f} q={r.get("Queue_Id","?")} {r["Kernel_Name"][:50]}')
P
tail -45 $R/gpurun_out/r03_trace_shstream.txt
