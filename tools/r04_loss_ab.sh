# GPU box: parity of the loss kernels in the tree, then A/B of the full iteration against main's library (exp/libs/main)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
val() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d.get('kernels', {})
        print(d['value'], 'l1_ssim_fwd', k.get('l1_ssim_fwd', {}).get('avg_us'), 'l1_ssim_bwd', k.get('l1_ssim_bwd', {}).get('avg_us'))
"; }
timeout 200 python -m pytest tests/test_gpu_loss.py -x -q 2>&1 | tail -5
B="python bench.py --no-cpu-baseline --loss l1_ssim --optimizer fused_adam --profile-steps 10 --steps 100 --warmup 10"
cp gaussian-mesh-splatting_amd/lib/libgmsplat.so /tmp/new.so
cp exp/libs/main/libgmsplat.so gaussian-mesh-splatting_amd/lib/libgmsplat.so
echo "main library   : $(timeout 120 $B 2>/dev/null | val)"
cp /tmp/new.so gaussian-mesh-splatting_amd/lib/libgmsplat.so
echo "tree's library : $(timeout 120 $B 2>/dev/null | val)"
echo "tree's library : $(timeout 120 $B --workload c5_flame_like_500k --steps 40 2>/dev/null | val)  (1024x1024)"
