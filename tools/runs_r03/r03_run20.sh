cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for s in 5113 5179 5312 5337 5364 5526 5838; do
  for M in 0 1; do
    echo "== seed $s GMS_MICRO=$M"; GMS_MICRO=$M timeout 120 python tools/fuzz_parity.py 1 $s 2>&1 | grep -E "^ok|MISMATCH|worst" | sed 's/max_clean.*amb_frac[^}]*}//' | cut -c1-260
  done
done > gpurun_out/r03_fuzz_bad7_compare.txt 2>&1
cat gpurun_out/r03_fuzz_bad7_compare.txt
timeout 600 python -m pytest tests/test_gpu_c4.py -q -x 2>&1 | tail -3
GMS_BENCH_FORCE_DDP=1 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --workload c4_ficus_like --profile-steps 0 > gpurun_out/r03_bench_i_one_rank.log 2>&1
python - <<'P'
import json
d=json.loads([l for l in open("gpurun_out/r03_bench_i_one_rank.log") if l.startswith("{")][-1])
print(d["value"], d["no_comm"], d["sh_exchange_ms_per_step"], d["sh_exchange"][:30], d["exchange_bytes"])
P
