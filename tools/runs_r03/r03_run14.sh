set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r03_bench_d_$tag.log 2> gpurun_out/r03_bench_d_$tag.err; }
run s0 GMS_BWD_SPLIT=0
run s3 GMS_BWD_SPLIT=3
run s5 GMS_BWD_SPLIT=5
run s6 GMS_BWD_SPLIT=6
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_bench_d_*.log")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, d["value"], {k:v["avg_us"] for k,v in d["kernels"].items() if k.startswith(("blend_bwd"))})
    except Exception as e: print(f, "ERR", e)
P
GMS_BWD_SPLIT=5 timeout 300 python -m pytest tests/test_gpu_raster.py -q -x -k "forward_backward_parity or full_size or repeated" 2>&1 | tail -2
