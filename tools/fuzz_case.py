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
    rng = np.random.default_rng(seed0 + case)
    P = int(rng.choice([1, 7, 100, 1500, 6000, 20000]))
    W, H = int(rng.integers(17, 300)), int(rng.integers(17, 220))
    deg = int(rng.integers(0, 4)); aa = bool(rng.integers(0, 2))
    lo = float(rng.choice([0.002, 0.01, 0.05])); hi = lo * float(rng.choice([2, 10, 40]))
    op_lo = float(rng.choice([0.01, 0.1, 0.6])); op_hi = min(0.999, op_lo + float(rng.choice([0.05, 0.4])))
    sc = syn.random_scene(P, seed=seed0 + case, scale_lo=lo, scale_hi=hi, opacity_lo=op_lo, opacity_hi=op_hi)
    cam = syn.orbit_camera(int(rng.integers(0, 8)), width=W, height=H, radius=float(rng.choice([1.5, 3.0, 6.0])))
    bg = torch.tensor(rng.uniform(0, 1, 3), dtype=torch.float32)
    kw = U.settings_kwargs(cam, bg, antialiasing=aa, sh_degree=deg, scale_modifier=float(rng.choice([1.0, 0.6, 1.8])))
    inputs = dict(means3D=sc.means3D, opacities=sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    o32 = U.oracle_render(inputs, kw)
    gc = syn.upstream_grad(torch.from_numpy(o32["color"])).numpy() * 1000.0
    gd = np.full((1, H, W), 1e-3, np.float32) if rng.integers(0, 2) else None
    o32 = U.oracle_render(inputs, kw, gc, gd)
    o64 = U.oracle_render(inputs, kw, gc, gd, precision="f64")
    h = U.hip_render(inputs, kw, grad_color=gc, grad_invdepth=gd)
    print(f"case {case}: P={P} {W}x{H} deg={deg} aa={aa}")
    for k in ("means3D", "means2D", "scales", "rotations", "opacities", "shs"):
        ref = np.asarray(o64["grads"][k], np.float64)
        if ref is None or ref.size == 0: continue
        den = np.abs(ref) + 1e-3 * np.abs(ref).max() + 1e-30
        eh = (np.abs(np.asarray(h["grads"][k], np.float64) - ref) / den).max()
        eo = (np.abs(np.asarray(o32["grads"][k], np.float64) - ref) / den).max()
        print(f"   {k:10s} max rel err vs f64:  hip {eh:.2e}   oracle-f32 {eo:.2e}")
