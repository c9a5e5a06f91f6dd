# One call (GPU box), branch next/loss-streaming: parity of the column-streaming loss kernels, A/B against main's library
# (exp/libs/main), and -- only if both are good -- the evidence of the new build (PMC passes, bench lines, config 3, the GPU tests
# that touch the loss).  Every step has its own timeout; outputs land in gpurun_out/ as they are produced.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=r04b
val() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d.get('kernels', {})
        print(d['value'], 'l1_ssim_fwd', k.get('l1_ssim_fwd', {}).get('avg_us'), 'l1_ssim_bwd', k.get('l1_ssim_bwd', {}).get('avg_us'))
"; }
timeout 200 python -m pytest tests/test_gpu_loss.py -x -q 2>&1 | tail -5 > gpurun_out/${T}_pytest_loss.log
cat gpurun_out/${T}_pytest_loss.log
grep -q " passed" gpurun_out/${T}_pytest_loss.log && ! grep -q "failed\|error" gpurun_out/${T}_pytest_loss.log || { echo "LOSS PARITY FAILED: stop"; exit 0; }
B="python bench.py --no-cpu-baseline --loss l1_ssim --optimizer fused_adam --profile-steps 10 --steps 100 --warmup 10"
cp gaussian-mesh-splatting_amd/lib/libgmsplat.so /tmp/new.so
cp exp/libs/main/libgmsplat.so gaussian-mesh-splatting_amd/lib/libgmsplat.so
OLD=$(timeout 120 $B 2>/dev/null | val); echo "main library   : $OLD"
cp /tmp/new.so gaussian-mesh-splatting_amd/lib/libgmsplat.so
NEW=$(timeout 120 $B 2>/dev/null | val); echo "streaming loss : $NEW"
(echo "full iteration (fwd+bwd + fused L1+SSIM + FusedAdam), 100 steps, it/s and the loss kernels' HIP-event us"; echo "main library   : $OLD"; echo "streaming loss : $NEW") > gpurun_out/${T}_loss_ab.txt
python -c "
import sys
o, n = float('$OLD'.split()[0]), float('$NEW'.split()[0])
sys.exit(0 if n > o * 1.01 else 1)" || { echo "NOT FASTER: stop"; exit 0; }
# ---- evidence of this build
SKIP_MARKER=1 bash tools/collect_profiles.sh $T > gpurun_out/${T}_collect.log 2>&1
python tools/make_pmc_traffic.py gpurun_out/${T}_rocprofv3_summary_traffic.json c2_hotdog_like/trained - profiles/r02_blend_stats_c2.json > gpurun_out/${T}_make_pmc.log 2>&1
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
timeout 150 python bench.py --steps 200 --warmup 20 > gpurun_out/${T}_bench_full.json.log 2> gpurun_out/${T}_bench_full.err
timeout 100 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --loss l1_ssim --optimizer fused_adam > gpurun_out/${T}_bench_full_iteration.json.log 2>&1
timeout 100 python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver_args.json.log 2> /dev/null
timeout 150 python tools/train_c3.py > gpurun_out/${T}_train_c3_7000iters.json.log 2> gpurun_out/${T}_train_c3.err
grep -h -o '"value": [0-9.]*' gpurun_out/${T}_bench_*.json.log; grep -h -o '"iters_per_s": [0-9.]*\|"psnr_mean_after": [0-9.]*' gpurun_out/${T}_train_c3_*.json.log
timeout 200 python -m pytest tests/test_gpu_loss.py tests/test_gpu_training.py tests/test_gpu_bench.py tests/test_gpu_optim.py -q 2>&1 | tail -4 > gpurun_out/${T}_pytest_gpu_loss_users.log; cat gpurun_out/${T}_pytest_gpu_loss_users.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; tail -2 gpurun_out/${T}_smoke.log
LEFT=$((470 - SECONDS)); echo "seconds used: $SECONDS"
if [ $LEFT -gt 60 ]; then timeout $LEFT python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > gpurun_out/${T}_pytest_gpu.log; cat gpurun_out/${T}_pytest_gpu.log; fi
