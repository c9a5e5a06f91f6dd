set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export GMS_MICRO=1
export GMS_SEG_LEN=256
timeout 600 python -m pytest tests/test_gpu_raster.py tests/test_gpu_negative_controls.py -q -x 2>&1 | tail -5 > gpurun_out/r03_pytest_micro6_a.log
tail -3 gpurun_out/r03_pytest_micro6_a.log
for F in 0 9 10 11; do
  GMS_FAULT=$F timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r03_bench_micro6_F$F.log 2> gpurun_out/r03_bench_micro6_F$F.err
done
for L in 128 512; do
  GMS_SEG_LEN=$L timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r03_bench_micro6_L$L.log 2> gpurun_out/r03_bench_micro6_L$L.err
done
GMS_TRIP=4 GMS_TRIP_BWD=1 timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r03_bench_micro6_t41.log 2>&1
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_bench_micro6_*.log")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, d["value"], {k:v["avg_us"] for k,v in d["kernels"].items() if k.startswith(("blend","micro"))})
    except Exception as e: print(f, "ERR", e)
P
