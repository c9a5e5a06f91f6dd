"""Re-run chosen fuzz cases against the float64 oracle: errors of the HIP path and of the float32 oracle, side by side."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (ROOT, os.path.join(ROOT, "gaussian-mesh-splatting_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import _util as U
from games_hip import synthetic as syn
seed0 = int(sys.argv[1]); cases = [int(c) for c in sys.argv[2:]]
for case in cases:
    inputs, kw, tag, rng = U.fuzz_case(seed0 + case)       # the sweep's generator (tests/_util.py)
    W, H = kw["image_width"], kw["image_height"]
    o32 = U.oracle_render(inputs, kw)
    gc = syn.upstream_grad(torch.from_numpy(o32["color"])).numpy() * 1000.0
    gd = np.full((1, H, W), 1e-3, np.float32) if rng.integers(0, 2) else None
    o32 = U.oracle_render(inputs, kw, gc, gd)
    o64 = U.oracle_render(inputs, kw, gc, gd, precision="f64")
    h = U.hip_render(inputs, kw, grad_color=gc, grad_invdepth=gd)
    print(f"case {case}: {tag}")
    for k in ("means3D", "means2D", "scales", "rotations", "opacities", "shs"):
        ref = np.asarray(o64["grads"][k], np.float64)
        if ref is None or ref.size == 0: continue
        den = np.abs(ref) + 1e-3 * np.abs(ref).max() + 1e-30
        eh = (np.abs(np.asarray(h["grads"][k], np.float64) - ref) / den).max()
        eo = (np.abs(np.asarray(o32["grads"][k], np.float64) - ref) / den).max()
        print(f"   {k:10s} max rel err vs f64:  hip {eh:.2e}   oracle-f32 {eo:.2e}")
