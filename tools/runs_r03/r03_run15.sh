set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_negative_controls.py tests/test_gpu_mesh.py -q -x 2>&1 | tail -6 > gpurun_out/r03_pytest_e.log
tail -3 gpurun_out/r03_pytest_e.log
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r03_bench_e_$tag.log 2> gpurun_out/r03_bench_e_$tag.err; }
run base X=1
run L128 GMS_SEG_LEN=128
run L192 GMS_SEG_LEN=192
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_bench_e_*.log")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], {k:v["avg_us"] for k,v in d["kernels"].items() if k.startswith(("blend","micro"))})
    except Exception as e: print(f, "ERR", e)
P
