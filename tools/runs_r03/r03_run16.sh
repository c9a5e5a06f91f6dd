set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_raster.py -q -x 2>&1 | tail -6 > gpurun_out/r03_pytest_g.log
tail -3 gpurun_out/r03_pytest_g.log
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r03_bench_g_$tag.log 2> gpurun_out/r03_bench_g_$tag.err; }
run reg X=1
run lds GMS_PRESORT=lds
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_bench_g_*.log")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], {k:(v["avg_us"],v["launches_per_step"]) for k,v in d["kernels"].items() if k.startswith(("tile","emit"))})
    except Exception as e: print(f, "ERR", e)
P
timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --workload c5_flame_like_1m --profile-steps 10 > gpurun_out/r03_c5g.log 2>&1
grep -h -o '"value": [0-9.]*' gpurun_out/r03_c5g.log
