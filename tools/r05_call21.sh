# round 5, GPU call 21: which HSA queue / stream each kernel of the headline step runs on (columns of the rocprofv3 kernel trace)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; T=r05u
rm -rf /tmp/prof_q
timeout 90 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_q -o t -- python $R/bench.py --no-cpu-baseline --profile-steps 0 --steps 60 --warmup 20 > /dev/null 2>&1
F=$(find /tmp/prof_q -name "*kernel_trace.csv" | head -1)
python - > $R/gpurun_out/${T}_kernel_queues.txt 2>&1 <<PY
import csv, collections
rows = list(csv.DictReader(open("$F")))
print("columns:", list(rows[0].keys()))
rows = rows[len(rows) // 2:]
c = collections.Counter((r["Kernel_Name"].split("(")[0][:50], r.get("Queue_Id"), r.get("Stream_Id"), r.get("Workgroup_Size"), r.get("Grid_Size"), r.get("LDS_Block_Size"), r.get("Scratch_Size"), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count")) for r in rows)
for k, n in sorted(c.items(), key=lambda kv: -kv[1])[:30]:
    print(n, k)
PY
cut -c1-220 $R/gpurun_out/${T}_kernel_queues.txt | head -24
