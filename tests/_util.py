"""Helpers shared by the GPU parity tests: run the HIP path and the CPU oracle on the same
inputs and compare with the discontinuity-aware rules described in DESIGN.md (parity section)."""
import math

import numpy as np
import torch

from oracle import gs_oracle


def settings_kwargs(cam, bg, sh_degree=3, antialiasing=False, scale_modifier=1.0):
    return dict(image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                bg=bg, scale_modifier=scale_modifier, viewmatrix=cam.world_view_transform,
                projmatrix=cam.full_proj_transform, sh_degree=sh_degree, campos=cam.camera_center,
                prefiltered=False, debug=False, antialiasing=antialiasing)


def hip_render(inputs: dict, kw: dict, device="cuda", grad_color=None, grad_invdepth=None, need_grad=True):
    """inputs: means3D, opacities and (shs | colors_precomp), (scales, rotations | cov3D_precomp) as CPU tensors."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device(device)
    t = {k: (v.to(dev).float().detach().clone().requires_grad_(need_grad) if v is not None else None)
         for k, v in inputs.items()}
    kwd = dict(kw)
    for k in ("bg", "viewmatrix", "projmatrix", "campos"):
        kwd[k] = kwd[k].to(dev).float()
    rs = GaussianRasterizationSettings(**kwd)
    means2D = torch.zeros_like(t["means3D"], requires_grad=need_grad)
    color, radii, invd = GaussianRasterizer(rs)(
        means3D=t["means3D"], means2D=means2D, opacities=t["opacities"], shs=t.get("shs"),
        colors_precomp=t.get("colors_precomp"), scales=t.get("scales"), rotations=t.get("rotations"),
        cov3D_precomp=t.get("cov3D_precomp"))
    out = dict(color=color.detach().cpu().numpy(), radii=radii.cpu().numpy(), invdepth=invd.detach().cpu().numpy())
    if grad_color is not None:
        loss = (color * torch.as_tensor(grad_color, device=dev, dtype=torch.float32)).sum()
        if grad_invdepth is not None:
            loss = loss + (invd * torch.as_tensor(grad_invdepth, device=dev, dtype=torch.float32)).sum()
        loss.backward()
        g = {k: (v.grad.detach().cpu().numpy() if v is not None and v.grad is not None else None) for k, v in t.items()}
        g["means2D"] = means2D.grad.detach().cpu().numpy()
        out["grads"] = g
    torch.cuda.synchronize()
    return out


def oracle_render(inputs: dict, kw: dict, grad_color=None, grad_invdepth=None, precision="f32"):
    okw = {k: v for k, v in kw.items() if k not in ("prefiltered", "debug")}
    o = gs_oracle.rasterize(**{k: v for k, v in inputs.items() if v is not None}, **okw, precision=precision)
    res = dict(color=o.color, radii=o.radii, invdepth=o.invdepth, N=o.N, interactions=o.interactions,
               details=o.state.details())
    if grad_color is not None:
        g = gs_oracle.backward(o, grad_color, grad_invdepth)
        res["grads"] = dict(means3D=g["means3D"], means2D=g["means2D"], shs=g["sh"], colors_precomp=g["colors_precomp"],
                            opacities=g["opacities"], scales=g["scales"], rotations=g["rotations"],
                            cov3D_precomp=g["cov3D_precomp"])
    return res


def forward_report(hip, ora, W, H):
    """Discontinuity-aware comparison.  Returns dict of statistics."""
    d = ora["details"]
    mism = hip["radii"] != ora["radii"]
    unexplained = mism & ~(d["gauss_ambig"].astype(bool))
    amb = d["pix_ambig"].astype(bool).copy()
    gx, gy = (W + 15) // 16, (H + 15) // 16
    # pixels of tiles touched by Gaussians whose discrete footprint differs are excluded too
    for i in np.nonzero(mism)[0]:
        r = max(int(hip["radii"][i]), int(ora["radii"][i]))
        x, y = d["xy"][i]
        x0, x1 = max(0, int((x - r) // 16) - 1), min(gx, int((x + r) // 16) + 2)
        y0, y1 = max(0, int((y - r) // 16) - 1), min(gy, int((y + r) // 16) + 2)
        amb[y0 * 16:y1 * 16, x0 * 16:x1 * 16] = True
    diff = np.abs(hip["color"] - ora["color"]).max(axis=0)
    ddiff = np.abs(hip["invdepth"][0] - ora["invdepth"][0])
    clean = ~amb
    mse = float(np.mean((hip["color"] - ora["color"]) ** 2))
    return dict(radii_mismatch=int(mism.sum()), radii_unexplained=int(unexplained.sum()), amb_frac=float(amb.mean()),
                max_clean=float(diff[clean].max()) if clean.any() else 0.0,
                max_amb=float(diff[amb].max()) if amb.any() else 0.0,
                max_invdepth_clean=float(ddiff[clean].max()) if clean.any() else 0.0,
                psnr=float(10 * math.log10(1.0 / mse)) if mse > 0 else float("inf"))


def grad_report(gh, go, q=0.999):
    """Per-tensor error statistics relative to the tensor's own scale."""
    rep = {}
    for k, b in go.items():
        a = gh.get(k)
        if a is None or b is None or b.size == 0:
            continue
        a = a.reshape(b.shape)
        scale = float(np.abs(b).max())
        if scale == 0:
            rep[k] = dict(scale=0.0, max_abs=float(np.abs(a).max()), q_rel=0.0, max_rel=0.0)
            continue
        err = np.abs(a - b)
        rel = err / (np.abs(b) + 1e-3 * scale)      # 1e-3 relative with an absolute floor of 1e-3*max|g|
        rep[k] = dict(scale=scale, max_abs=float(err.max()), q_rel=float(np.quantile(rel, q)), max_rel=float(rel.max()),
                      frac_bad=float((rel > 1e-3).mean()))
    return rep
