cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r03_bench_h_$tag.log 2> gpurun_out/r03_bench_h_$tag.err; }
run ur1 GMS_UNIT_RUN=1
run ur4 GMS_UNIT_RUN=4
run ur16 GMS_UNIT_RUN=16
run ur64 GMS_UNIT_RUN=64
run t4 GMS_TRIP=4
run t4b4 GMS_TRIP=4 GMS_TRIP_BWD=4
run t1 GMS_TRIP=1
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_bench_h_*.log")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, d["value"], {k:v["avg_us"] for k,v in d["kernels"].items() if k.startswith(("blend"))})
    except Exception as e: print(f, "ERR", e)
P
