set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export GMS_MICRO=1
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r03_bench_micro11_$tag.log 2> gpurun_out/r03_bench_micro11_$tag.err; }
run rot GMS_SEG_LEN=256
run norot GMS_SEG_LEN=256 GMS_MICRO_NOROT=1
run quadrot GMS_SEG_LEN=256 GMS_MICRO_ORDER=quad
run rotF9 GMS_SEG_LEN=256 GMS_FAULT=9
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_bench_micro11_*.log")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, d["value"], {k:v["avg_us"] for k,v in d["kernels"].items() if k.startswith(("blend","micro"))})
    except Exception as e: print(f, "ERR", e)
P
timeout 300 python -m pytest tests/test_gpu_raster.py -q -x -k "forward_backward_parity or full_size" 2>&1 | tail -2
