set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export GMS_MICRO=1
export GMS_SEG_LEN=256
timeout 300 python -m pytest tests/test_gpu_raster.py -q -x -k "forward_backward_parity or full_size or long_tile or config5" 2>&1 | tail -5 > gpurun_out/r03_pytest_micro3_a.log
tail -3 gpurun_out/r03_pytest_micro3_a.log
for F in 0 9 10 11; do
  GMS_FAULT=$F timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r03_bench_micro3_F$F.log 2> gpurun_out/r03_bench_micro3_F$F.err
done
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_bench_micro3_F*.log")):
    d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f, d["value"], {k:v["avg_us"] for k,v in d["kernels"].items() if k.startswith(("blend","micro"))})
P
