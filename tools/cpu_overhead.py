"""Host-side (enqueue) time of each part of a training iteration, no device syncs inside the loop."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-mesh-splatting_amd"))
import torch
from games_hip import synthetic as syn
from games_hip.model import HipGaussianMeshModel
from games_hip.render import PipelineParams, render
from games_hip.loss import l1_ssim_loss

dev = torch.device("cuda", 0)
scene = syn.mesh_scene("c2_hotdog_like", state="trained")
size = scene.meta["image"]
model = HipGaussianMeshModel.from_scene(scene, dev)
cam = syn.orbit_camera(0, width=size, height=size).to(dev)
bg = torch.ones(3, device=dev); pipe = PipelineParams()
gt = torch.rand(3, size, size, device=dev)
model.training_setup(1e-12, 1e-12, 1e-12, 1e-12, 1e-12, fused=sys.argv[1:] != ["torch"])
acc = {}
def tick(name, t0):
    t = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t - t0); return t
import ctypes as C
from diff_gaussian_rasterization import _lib
lib = _lib.load()
for it in range(220):
    if it == 20:
        torch.cuda.synchronize(); acc.clear(); wall0 = time.perf_counter(); lib.gms_wait_stats(None, None, 1)
    t = time.perf_counter()
    model.update_alpha(); model.prepare_scaling_rot(); t = tick("k0_fwd", t)
    image = render(cam, model, pipe, bg)["render"]; t = tick("render_fwd", t)
    loss = l1_ssim_loss(image, gt, 0.2); t = tick("loss_fwd", t)
    loss.backward(); t = tick("backward", t)
    model.optimizer.step(); t = tick("optimizer.step", t)
    model.optimizer.zero_grad(set_to_none=True); t = tick("zero_grad", t)
host = time.perf_counter() - wall0
torch.cuda.synchronize()
wall = time.perf_counter() - wall0
ms, n = C.c_double(0), C.c_int64(0); lib.gms_wait_stats(C.byref(ms), C.byref(n), 0)
print("wait for N inside forward: %.1f us/iter" % (ms.value * 1e3 / max(n.value, 1)))
print({k: round(v / 200 * 1e6, 1) for k, v in acc.items()}, "host us/iter", round(host / 200 * 1e6, 1), "wall us/iter", round(wall / 200 * 1e6, 1))
