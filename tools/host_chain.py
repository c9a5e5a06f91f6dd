"""Host-side time of each part of the HEADLINE step (K0 fwd + render fwd + bwd, bench.py's make_step), no device syncs inside the
loop, and the time the forward spends waiting for the frame's instance count (gms_wait_stats): host work per step = host - wait.
If the wait is ~0 the step is host-bound on this box.   usage: python tools/host_chain.py [steps]"""
import ctypes as C
import os
import sys
import time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-mesh-splatting_amd"))
import torch
from games_hip import synthetic as syn
from games_hip.model import HipGaussianMeshModel
from games_hip.render import PipelineParams, render
from diff_gaussian_rasterization import _lib

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda", 0)
scene = syn.mesh_scene("c2_hotdog_like", state="trained")
size = scene.meta["image"]
model = HipGaussianMeshModel.from_scene(scene, dev)
cam = syn.orbit_camera(0, width=size, height=size).to(dev)
bg = torch.ones(3, device=dev); pipe = PipelineParams()
params = [p for p in (model._alpha, model._scale, model._opacity, model._features_dc, model._features_rest, model.vertices) if p.requires_grad]
inv_norm = 1.0 / (3.0 * size * size)
neg_half = torch.tensor(-0.5 * inv_norm, device=dev)
lib = _lib.load()
acc = {}


def tick(name, t0):
    t = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t - t0); return t


for it in range(steps + 50):
    if it == 50:
        torch.cuda.synchronize(); acc.clear(); wall0 = time.perf_counter(); lib.gms_wait_stats(None, None, 1)
    t = time.perf_counter()
    model.update_alpha(); model.prepare_scaling_rot(); t = tick("k0_fwd", t)
    image = render(cam, model, pipe, bg)["render"]; t = tick("render_fwd (incl. the wait for N)", t)
    with torch.no_grad():
        g = torch.add(neg_half, image, alpha=inv_norm)
    t = tick("upstream gradient", t)
    torch.autograd.backward([image], [g]); t = tick("backward", t)
    for p in params:
        p.grad = None
    t = tick("reset grads", t)
host = time.perf_counter() - wall0
torch.cuda.synchronize()
wall = time.perf_counter() - wall0
ms, n = C.c_double(0), C.c_int64(0); lib.gms_wait_stats(C.byref(ms), C.byref(n), 0)
wait = ms.value * 1e3 / max(n.value, 1)
print({k: round(v / steps * 1e6, 1) for k, v in acc.items()})
print("host us/step %.1f   of which waiting for N %.1f   => host work %.1f   wall us/step %.1f (%.0f it/s)"
      % (host / steps * 1e6, wait, host / steps * 1e6 - wait, wall / steps * 1e6, steps / wall))
