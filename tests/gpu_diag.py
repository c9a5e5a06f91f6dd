"""Diagnostic (not a test): prints HIP-vs-oracle error statistics.  Run on the GPU box."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "gaussian-mesh-splatting_amd")
sys.path.insert(0, "tests")
import _util as U  # noqa: E402
from games_hip import synthetic as syn  # noqa: E402


def run(name, sc, cam, bg, **kw):
    inputs = dict(means3D=sc.means3D, opacities=sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    k = U.settings_kwargs(cam, bg, **kw)
    W, H = cam.image_width, cam.image_height
    o = U.oracle_render(inputs, k)
    gc = syn.upstream_grad(torch.from_numpy(o["color"])).numpy() * 1000.0
    gd = np.full((1, H, W), 1e-3, np.float32)
    t0 = time.time(); o = U.oracle_render(inputs, k, gc, gd); t_or = time.time() - t0
    h = U.hip_render(inputs, k, grad_color=gc, grad_invdepth=gd)
    print(f"== {name}: P={sc.means3D.shape[0]} {W}x{H} N={o['N']} inter={o['interactions']:.3g} oracle {t_or:.2f}s")
    print("  fwd:", json.dumps(U.forward_report(h, o, W, H)))
    for kk, v in U.grad_report(h["grads"], o["grads"]).items():
        print(f"  grad {kk:14s}", json.dumps(v))


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    run("random-2k", syn.random_scene(2000, seed=1, scale_lo=0.01, scale_hi=0.1), syn.orbit_camera(1, width=160, height=128, radius=3.0), torch.tensor([0.2, 0.4, 0.6]))
    run("random-aa", syn.random_scene(3000, seed=2, scale_lo=0.005, scale_hi=0.08), syn.orbit_camera(2, width=200, height=120, radius=3.0), torch.ones(3), antialiasing=True, sh_degree=2)
    run("flat-10k", syn.flat_scene(10000), syn.orbit_camera(0, width=256, height=256), torch.ones(3))
