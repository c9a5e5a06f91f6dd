# Round-6 evidence run (GPU box): rocprofv3 + PMC passes first (so the bench lines carry counters of THIS build), the VALU model, headline
# and secondary bench lines, config 3 (default and deterministic), fuzz sweeps, phase stamps, full GPU suite, smoke.   bash tools/r06_final.sh [TAG]
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-r06}
bash tools/collect_profiles.sh $T > gpurun_out/${T}_collect.log 2>&1
python tools/make_pmc_traffic.py gpurun_out/${T}_rocprofv3_summary_traffic.json c2_hotdog_like/trained - profiles/r02_blend_stats_c2.json > gpurun_out/${T}_make_pmc.log 2>&1
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
python tools/valu_mix.py > profiles/valu_model.json 2> gpurun_out/${T}_valu_mix.err; cp profiles/valu_model.json gpurun_out/valu_model.json
tools/valu_bench.bin > gpurun_out/${T}_valu_microbenchmark.txt 2>&1
python bench.py --steps 200 --warmup 20 > gpurun_out/${T}_bench_full.json.log 2> gpurun_out/${T}_bench_full.err
python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver_args.json.log 2> /dev/null
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-fused-k0 > gpurun_out/${T}_bench_k0_own_launch.json.log 2>&1
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --loss l1_ssim --optimizer fused_adam > gpurun_out/${T}_bench_full_iteration.json.log 2>&1
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --workload c4_ficus_like > gpurun_out/${T}_bench_c4_single_gpu.json.log 2>&1
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --workload c5_flame_like_1m --profile-steps 10 > gpurun_out/${T}_bench_c5_1m.json.log 2>&1
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --workload c5_flame_like_1m --mode animate > gpurun_out/${T}_bench_c5_1m_animate.json.log 2>&1
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --workload c5_flame_like_500k --profile-steps 10 > gpurun_out/${T}_bench_c5_500k.json.log 2>&1
python bench.py --steps 100 --warmup 20 --no-cpu-baseline --workload c5_flame_like_500k --mode animate > gpurun_out/${T}_bench_c5_500k_animate.json.log 2>&1
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --workload small --mode animate > gpurun_out/${T}_bench_small_animate.json.log 2>&1
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --workload small --mode animate --graph > gpurun_out/${T}_bench_small_animate_graph.json.log 2>&1
GMS_BENCH_FORCE_DDP=1 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --workload c4_ficus_like --profile-steps 0 > gpurun_out/${T}_bench_one_rank_rccl.json.log 2>&1
python bench.py --gpus 2 --steps 50 --warmup 10 --no-cpu-baseline --profile-steps 0 > gpurun_out/${T}_bench_gpus2_shared.json.log 2>&1
timeout 900 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 > gpurun_out/${T}_bench_gpus8_shared.json.log 2>&1
GAMES_HIP_DETERMINISTIC=1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --profile-steps 10 > gpurun_out/${T}_bench_deterministic.json.log 2>&1
python tools/train_c3.py > gpurun_out/${T}_train_c3_7000iters.json.log 2> gpurun_out/${T}_train_c3.err
GAMES_HIP_DETERMINISTIC=1 python tools/train_c3.py > gpurun_out/${T}_train_c3_7000iters_deterministic.json.log 2> gpurun_out/${T}_train_c3_det.err
grep -h -o '"value": [0-9.]*' gpurun_out/${T}_bench_*.json.log
grep -h -o '"iters_per_s": [0-9.]*' gpurun_out/${T}_train_c3_*.json.log
grep -h -o '"roofline": {[^}]*}' gpurun_out/${T}_bench_full.json.log | cut -c1-300
grep -E "^(micro|blend|tile_|preprocess|emit|mesh)" gpurun_out/${T}_rocprofv3_summary.txt | head -14 | cut -c1-130
bash tools/r06_call.sh phases $T
bash tools/r06_call.sh fwdphases $T
python tools/fuzz_parity.py ${FUZZ_N:-300} 71000 > gpurun_out/${T}_fuzz_${FUZZ_N:-300}cases.log 2>&1; tail -4 gpurun_out/${T}_fuzz_${FUZZ_N:-300}cases.log | cut -c1-250
python tools/fuzz_parity.py ${FUZZ_DET_N:-100} 73000 det > gpurun_out/${T}_fuzz_${FUZZ_DET_N:-100}cases_deterministic_strict.log 2>&1; tail -4 gpurun_out/${T}_fuzz_${FUZZ_DET_N:-100}cases_deterministic_strict.log | cut -c1-250
rm -f gpurun_out/parity_report.jsonl
python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/${T}_pytest_gpu.log
tail -3 gpurun_out/${T}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; tail -2 gpurun_out/${T}_smoke.log
