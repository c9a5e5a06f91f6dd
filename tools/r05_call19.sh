# round 5, GPU call 19: which HIP API calls a headline step makes (rocprofv3 --hip-trace --stats; 100 timed + 20 warm-up steps + setup)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; T=r05s
rm -rf /tmp/prof_hip
timeout 100 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d /tmp/prof_hip -o t -- python $R/bench.py --no-cpu-baseline --profile-steps 0 --steps 100 --warmup 20 > $R/gpurun_out/${T}_bench_traced.json.log 2>/dev/null
(echo "# $(python -c "import json; j=json.loads(open('$R/gpurun_out/${T}_bench_traced.json.log').read().strip().splitlines()[-1]); print('traced (hip + kernel):', j['value'], 'it/s', j['ms_per_step'], 'ms')")"
 for f in $(find /tmp/prof_hip -name "*hip_api_stats.csv" | head -1); do echo "## $(basename $f)"; head -40 $f; done) > $R/gpurun_out/${T}_hip_api_stats.txt 2>&1
F=$(find /tmp/prof_hip -name "*kernel_trace.csv" | head -1)
python $R/tools/gpu_gaps.py $F 0.5 > $R/gpurun_out/${T}_gpu_gaps.txt 2>&1
python - > $R/gpurun_out/${T}_hip_calls_between_kernels.txt 2>&1 <<PY
import csv, glob, collections
api = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in csv.DictReader(open(glob.glob("/tmp/prof_hip/**/*hip_api_trace.csv", recursive=True)[0])))
ker = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]) for r in csv.DictReader(open("$F")))
ker = ker[len(ker) // 2:]
# for every idle gap > 2 us in front of a kernel: which HIP calls the host was inside during the gap
out = collections.defaultdict(lambda: collections.Counter())
n = collections.Counter()
import bisect
starts = [a[0] for a in api]
prev_end = ker[0][1]
for s, e, name in ker[1:]:
    if s - prev_end > 2000:
        n[name] += 1
        i = bisect.bisect_left(starts, prev_end - 200000)
        for a in api[i:]:
            if a[0] > s: break
            if a[1] >= prev_end: out[name][a[2]] += 1
    prev_end = max(prev_end, e)
for k in n:
    print(k, "gaps > 2 us:", n[k], dict(out[k].most_common(8)))
PY
head -12 $R/gpurun_out/${T}_hip_api_stats.txt | cut -c1-160; cat $R/gpurun_out/${T}_hip_calls_between_kernels.txt | cut -c1-300 | head
