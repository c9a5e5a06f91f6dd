set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r03_pytest_a.log
python bench.py --steps 200 --warmup 20 > gpurun_out/r03_bench_a.log 2> gpurun_out/r03_bench_a.err
GMS_BENCH_FORCE_DDP=1 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --workload c4_ficus_like --profile-steps 0 --sh-exchange factor > gpurun_out/r03_ddp1_factor.log 2>gpurun_out/r03_ddp1_factor.err
GMS_BENCH_FORCE_DDP=1 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --workload c4_ficus_like --profile-steps 0 --sh-exchange dense > gpurun_out/r03_ddp1_dense.log 2>gpurun_out/r03_ddp1_dense.err
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --workload c4_ficus_like --profile-steps 0 > gpurun_out/r03_c4_single.log 2>&1
timeout 300 python tools/fuzz_parity.py 400 > gpurun_out/r03_fuzz_400.log 2>&1
tail -3 gpurun_out/r03_pytest_a.log; tail -c 600 gpurun_out/r03_bench_a.log; tail -2 gpurun_out/r03_fuzz_400.log
