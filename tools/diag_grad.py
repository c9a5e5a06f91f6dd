"""GPU-box diagnostic: worst gradient entries (HIP vs f32 / f64 oracle) at c5 size on identical rasterizer inputs."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (ROOT, os.path.join(ROOT, "gaussian-mesh-splatting_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import _util as U
from games_hip import synthetic as syn
from oracle import mesh_oracle
wl = sys.argv[1] if len(sys.argv) > 1 else "c5_flame_like_1m"
scene = syn.mesh_scene(wl, state="trained")
size = scene.meta["image"]
cam = syn.orbit_camera(2, width=size, height=size)
with torch.no_grad():
    _, _, xyz, scaling, rot = mesh_oracle.mesh_to_gaussians(scene.vertices, scene.faces, scene._alpha, scene._scale)
    xa, sa, ra, oa, shs = mesh_oracle.activated(xyz, scaling, rot, scene._opacity, scene._features_dc, scene._features_rest)
inputs = dict(means3D=xa, opacities=oa, shs=shs, scales=sa, rotations=ra)
kw = U.settings_kwargs(cam, torch.ones(3))
o = U.oracle_render(inputs, kw)
gc = syn.upstream_grad(torch.from_numpy(o["color"])).numpy() * 1000.0
o = U.oracle_render(inputs, kw, gc, None)
o64 = U.oracle_render(inputs, kw, gc, None, precision="f64")
oacc = U.oracle_render(inputs, kw, gc, None, precision="f32acc")
h = U.hip_render(inputs, kw, grad_color=gc, grad_invdepth=None)
det = o["details"]
for k in ("scales", "rotations", "means3D", "means2D", "opacities"):
    a, b, t = h["grads"][k].astype(np.float64), o["grads"][k].astype(np.float64), o64["grads"][k].astype(np.float64)
    a = a.reshape(b.shape)
    sc = np.abs(b).max()
    eh, eo = np.abs(a - t), np.abs(b - t)
    eo = np.maximum(eo, np.abs(oacc["grads"][k].astype(np.float64).reshape(b.shape) - t))
    eo_row = eo.reshape(eo.shape[0], -1).max(1)
    ratio = (eh - 1e-3 * (np.abs(t) + 1e-3 * sc)).reshape(eh.shape[0], -1).max(1) / (eo_row + 1e-30)
    rel = (np.abs(a - b) / (np.abs(b) + 1e-3 * sc)).reshape(eh.shape[0], -1).max(1)
    idx = np.argsort(-np.where(rel > 1e-3, ratio, -1))[:4]
    idx = [i for i in idx if rel[i] > 1e-3]
    print(f"== {k}: scale {sc:.3e}, rows with rel>1e-3: {int((rel > 1e-3).sum())}")
    for i in idx:
        print(f"  g={i} ratio {ratio[i]:.1f} rel {rel[i]:.2e} hip {a[i].ravel()[:4]} o32 {b[i].ravel()[:4]} o64 {t[i].ravel()[:4]}")
        print(f"     radius {o['radii'][i]} scales {sa[i].numpy()} conic_op {det['conic_op'][i]} xy {det['xy'][i]} depth {det['depth'][i]:.3f} amb {det['gauss_ambig'][i]}")
        for kk in ("means2D", "opacities", "scales"):
            print(f"     {kk}: hip {h['grads'][kk][i].ravel()[:3]} o32 {o['grads'][kk][i].ravel()[:3]} acc {oacc['grads'][kk][i].ravel()[:3]} o64 {o64['grads'][kk][i].ravel()[:3]}")
        print(f"     rect {det['rect'][i]} rotation {ra[i].numpy()}")
