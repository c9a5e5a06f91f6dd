"""Diagnose one fuzz case (GPU box): where the forward images differ and which Gaussians carry the gradient outliers.
usage: diag_case.py <seed>   (seed = seed0 + case of tools/fuzz_parity.py)"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (ROOT, os.path.join(ROOT, "gaussian-mesh-splatting_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import _util as U
from games_hip import synthetic as syn
seed = int(sys.argv[1])
inputs, kw, tag, rng = U.fuzz_case(seed)       # the sweep's generator (tests/_util.py)
W, H = kw["image_width"], kw["image_height"]
P, deg, aa = inputs["means3D"].shape[0], kw["sh_degree"], kw["antialiasing"]
o = U.oracle_render(inputs, kw)
gc = syn.upstream_grad(torch.from_numpy(o["color"])).numpy() * 1000.0
gd = np.full((1, H, W), 1e-3, np.float32) if rng.integers(0, 2) else None
o = U.oracle_render(inputs, kw, gc, gd)
o64 = U.oracle_render(inputs, kw, gc, gd, precision="f64")
h = U.hip_render(inputs, kw, grad_color=gc, grad_invdepth=gd)
d = o["details"]
print(f"P={P} {W}x{H} deg={deg} aa={aa} N={o['N']}")
mism = np.nonzero(h["radii"] != o["radii"])[0]
print("radii mismatches:", [(int(i), int(h["radii"][i]), int(o["radii"][i]), int(d["gauss_ambig"][i])) for i in mism[:10]])
diff = np.abs(h["color"] - o["color"]).max(axis=0)
ys, xs = np.nonzero(diff > 1e-5)
print("pixels with |dC| > 1e-5:", [(int(x), int(y), float(diff[y, x]), int(d["pix_ambig"][y, x]), int(d["n_contrib"][y, x])) for x, y in zip(xs[:12], ys[:12])])
ref = np.asarray(o64["grads"]["means2D"], np.float64); a = np.asarray(h["grads"]["means2D"], np.float64); b = np.asarray(o["grads"]["means2D"], np.float64)
scale = np.abs(ref).max()
rel = np.abs(a - b) / (np.abs(b) + 1e-3 * np.abs(b).max())
rows = np.nonzero((rel > 1e-3).any(axis=1))[0]
print("means2D outlier rows:", len(rows), "scale", scale)
gx = (W + 15) // 16
for i in rows[:12]:
    x, y = d["xy"][i]
    print(f"  g{i}: xy=({x:.3f},{y:.3f}) depth={d['depth'][i]:.6f} radius={o['radii'][i]} amb={d['gauss_ambig'][i]} op={d['conic_op'][i][3]:.4f} conic={d['conic_op'][i][:3]}"
          f" hip={a[i][:2]} o32={b[i][:2]} o64={ref[i][:2]}")
# tiles of the differing pixels: entries around the depth of the outlier Gaussians
pl, rg = d["point_list"], d["ranges"]
for x, y in list(zip(xs, ys))[:3]:
    t = (y // 16) * gx + (x // 16)
    lst = pl[rg[t][0]:rg[t][1]]
    dep = d["depth"][lst]
    print(f"tile {t} of pixel ({x},{y}): {len(lst)} entries; outlier rows in it: {[int(g) for g in lst if g in set(rows.tolist())]}")
    dd = np.diff(dep)
    print("   equal-depth neighbours:", [(int(lst[k]), int(lst[k + 1]), float(dep[k])) for k in np.nonzero(dd == 0)[0][:8]],
          " min positive gap:", float(dd[dd > 0].min()) if (dd > 0).any() else None)
    # per-entry alpha at this pixel
    A, B, C, op = (d["conic_op"][lst, k] for k in range(4))
    dx, dy = d["xy"][lst, 0] - x, d["xy"][lst, 1] - y
    pw = -0.5 * (A * dx * dx + C * dy * dy) - B * dx * dy
    al = np.minimum(0.99, op * np.exp(pw))
    near = np.nonzero((np.abs(al * 255 - 1) < 1e-2) | (np.abs(pw) < 1e-4))[0]
    print("   entries near a threshold at this pixel:", [(int(lst[k]), float(al[k] * 255), float(pw[k])) for k in near[:8]])
    contrib = np.nonzero((pw <= 0) & (al >= 1 / 255))[0]
    print("   contributing entries (id, alpha):", [(int(lst[k]), round(float(al[k]), 4)) for k in contrib[:20]])
