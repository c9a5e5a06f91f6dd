"""Time the fused HIP L1+SSIM loss against the reference formulation (oracle restatement run with torch ops on the GPU)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-mesh-splatting_amd"))
import torch
from games_hip.loss import l1_ssim_loss
from oracle import loss_oracle

def bench(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

for hw in ((800, 800), (1024, 1024), (1080, 1920)):
    img = torch.rand(3, *hw, device="cuda", requires_grad=True); gt = torch.rand(3, *hw, device="cuda")
    def hip():
        img.grad = None; l1_ssim_loss(img, gt, 0.2).backward()
    def ref():
        img.grad = None; loss_oracle.l1_ssim_loss(img, gt, 0.2).backward()
    print(f"{hw}: hip fwd+bwd {bench(hip):.3f} ms   torch-ops (reference formulation) {bench(ref):.3f} ms", flush=True)
from diff_gaussian_rasterization import _lib
lib = _lib.load(); lib.gms_profile_enable(1); lib.gms_profile_reset()
img = torch.rand(3, 800, 800, device="cuda", requires_grad=True); gt = torch.rand(3, 800, 800, device="cuda")
for _ in range(20):
    img.grad = None; l1_ssim_loss(img, gt, 0.2).backward()
torch.cuda.synchronize()
print({k: round(v[0] / max(v[1], 1) * 1e3, 2) for k, v in _lib.kernel_times().items() if v[1]})
