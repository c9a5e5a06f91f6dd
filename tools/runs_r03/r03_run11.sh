set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_threads.py tests/test_gpu_negative_controls.py -q 2>&1 | tail -12 > gpurun_out/r03_pytest_b.log
tail -4 gpurun_out/r03_pytest_b.log
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r03_bench_b_$tag.log 2> gpurun_out/r03_bench_b_$tag.err; }
run sh1 GMS_SH_STREAM=1
run sh0 GMS_SH_STREAM=0
run c5 GMS_SH_STREAM=1 X=1 -- 
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_bench_b_*.log")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], {k:(v["avg_us"],v["launches_per_step"]) for k,v in d["kernels"].items()})
    except Exception as e: print(f, "ERR", e)
P
for M in 1 0; do
GMS_MICRO=$M timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --workload c5_flame_like_1m --profile-steps 10 > gpurun_out/r03_c5b_micro$M.log 2>&1
GMS_MICRO=$M timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --workload c5_flame_like_1m --mode animate > gpurun_out/r03_c5b_anim_micro$M.log 2>&1
done
grep -h -o '"value": [0-9.]*' gpurun_out/r03_c5b_*.log
