set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export GMS_MICRO=1
export GMS_SEG_LEN=256
for R in 0 1; do for F in 0 9 12; do
  GMS_MICRO_REGIONS=$R GMS_FAULT=$F timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r03_bench_micro7_R${R}_F$F.log 2> gpurun_out/r03_bench_micro7_R${R}_F$F.err
done; done
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_bench_micro7_*.log")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, d["value"], {k:v["avg_us"] for k,v in d["kernels"].items() if k.startswith(("blend","micro"))})
    except Exception as e: print(f, "ERR", e)
P
(rocprofv3 -L 2>/dev/null || rocprofv3 --list-avail 2>/dev/null) | grep -i -E "atomic|TCC_EA|TCC_REQ|TCC_HIT|TCC_MISS|TCP_TCC" | head -60 > gpurun_out/r03_counters_avail.txt
wc -l gpurun_out/r03_counters_avail.txt
