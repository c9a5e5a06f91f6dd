cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_driver_args.json.log 2> gpurun_out/r03_bench_driver_args.err; tail -c 600 gpurun_out/r03_bench_driver_args.err
python - <<'P'
import json
d=json.loads([l for l in open("gpurun_out/r03_bench_driver_args.json.log") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["traffic"], d["roofline"]["pmc"][:90], d["cpu_baseline"]["value"])
P
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
rm -f gpurun_out/parity_report.jsonl
python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r03_pytest_gpu_final.log; cat gpurun_out/r03_pytest_gpu_final.log
