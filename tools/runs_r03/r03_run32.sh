cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pr
GMS_BENCH_FORCE_DDP=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/pr/trace -o rr -- python $R/bench.py --no-cpu-baseline --profile-steps 0 --steps 60 --warmup 10 --workload c4_ficus_like --sh-exchange packed > /tmp/pr.log 2>&1
f=$(find /tmp/pr -name "*kernel_trace.csv" | head -1)
python $R/tools/gpu_gaps.py $f 0.85 > $R/gpurun_out/r03_gaps_one_rank_packed.txt 2>&1
head -22 $R/gpurun_out/r03_gaps_one_rank_packed.txt | cut -c1-120
python - "$f" <<'P'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
ev=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in rows)
ev=ev[int(len(ev)*0.85):]
d=collections.defaultdict(lambda:[0,0.0])
for s,e,n in ev:
    k=n.split("(")[0][:70]; d[k][0]+=1; d[k][1]+=(e-s)/1e3
nsteps=d[[k for k in d if "micro_bwd" in k][0]][0]
print("steps", nsteps)
for k,(n,t) in sorted(d.items(), key=lambda kv:-kv[1][1])[:24]: print(f"{k:72s} {n/nsteps:5.2f}/step {t/nsteps:8.2f} us/step")
P
