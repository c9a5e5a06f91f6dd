"""Dense float64 autograd formulation of the Gaussian rasterizer (second, independent checker).

TEST INFRASTRUCTURE ONLY (see oracle/gs_oracle.c).  Purpose: pin the hand-derived backward of
gs_oracle.c.  Everything here is written as plain differentiable torch ops over a dense
[pixels x Gaussians] matrix and differentiated by autograd; only the conventions in which the
rasterizer deliberately deviates from naive autograd are encoded:
  (i)   alpha = min(0.99, op*G) is straight-through,
  (ii)  cull / skip / stop tests are constants,
  (iv)  the t.x/t.z frustum clamp passes gradient only inside the limit and never to t.z,
  (x)   means2D receives d loss / d NDC (pixel-gradient * W/2, H/2).
  (vi)  the conic's backward divides by det^2 + 1e-7 instead of det^2 (`_RegularisedConic`: the three published lines;
        `regularised_conic=False` gives the exact derivative -- the two differ by up to 1e-7/det^2 relative, i.e. ~1e-5 for
        the smallest splats, whose det is ~0.09 after the 0.3 dilation).

In-tree sub-stages restated with a device/dtype argument because the originals hard-code
device="cuda": cov3D (scene/gaussian_model.py:27-31 + utils/general_utils.py:144-190) and
SH (utils/sh_utils.py:57-112).  O(P*H*W) memory: small scenes only.
"""
from __future__ import annotations


import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


class _RegularisedConic(torch.autograd.Function):
    """conic = (c, -b, a) / det with the published backward (SURVEY.md appendix A.5 vi): 1 / (det^2 + 1e-7) in place of
    1 / det^2.  The incoming gradient of the off-diagonal entry is the TRUE one here (the kernels store half of it)."""

    @staticmethod
    def forward(ctx, a, b, c):
        det = a * c - b * b
        ctx.save_for_backward(a, b, c, det)
        return c / det, -b / det, a / det

    @staticmethod
    def backward(ctx, gA, gB_true, gC):
        a, b, c, det = ctx.saved_tensors
        gB = 0.5 * gB_true
        d2 = 1.0 / (det * det + 1e-7)
        da = d2 * (-c * c * gA + 2 * b * c * gB + (det - a * c) * gC)
        dc = d2 * (-a * a * gC + 2 * a * b * gB + (det - a * c) * gA)
        db = d2 * 2 * (b * c * gA - (det + 2 * b * b) * gB + a * b * gC)
        return da, db, dc


def build_rotation_unnormalised(q):
    """utils/general_utils.py:158-179 without the normalisation (done in python by the caller)."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(-1, 3, 3)


def cov3d_python(scales, scale_modifier, rotations, normalise=False):
    """scene/gaussian_model.py:27-31 (`--compute_cov3D_python`), device-agnostic restatement.
    Returns the 6 packed upper-triangular entries (utils/general_utils.py:144-156)."""
    q = rotations / rotations.norm(dim=1, keepdim=True) if normalise else rotations
    R = build_rotation_unnormalised(q)
    L = R * (scale_modifier * scales)[:, None, :]
    S = L @ L.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=-1)


def eval_sh(deg, sh, dirs):
    """utils/sh_utils.py:57-112 restated; sh [..., C, K], dirs [..., 3]."""
    result = SH_C0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        result = result - SH_C1 * y * sh[..., 1] + SH_C1 * z * sh[..., 2] - SH_C1 * x * sh[..., 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            result = (result + SH_C2[0] * xy * sh[..., 4] + SH_C2[1] * yz * sh[..., 5]
                      + SH_C2[2] * (2.0 * zz - xx - yy) * sh[..., 6] + SH_C2[3] * xz * sh[..., 7]
                      + SH_C2[4] * (xx - yy) * sh[..., 8])
            if deg > 2:
                result = (result + SH_C3[0] * y * (3 * xx - yy) * sh[..., 9] + SH_C3[1] * xy * z * sh[..., 10]
                          + SH_C3[2] * y * (4 * zz - xx - yy) * sh[..., 11]
                          + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12]
                          + SH_C3[4] * x * (4 * zz - xx - yy) * sh[..., 13]
                          + SH_C3[5] * z * (xx - yy) * sh[..., 14] + SH_C3[6] * x * (xx - 3 * yy) * sh[..., 15])
    return result


def sh_colors_python(sh_degree, shs, means3D, campos):
    """renderer/gaussian_renderer/__init__.py:83-87 (`--convert_SHs_python`)."""
    shs_view = shs.transpose(1, 2)
    d = means3D - campos[None, :]
    d = d / d.norm(dim=1, keepdim=True)
    return torch.clamp_min(eval_sh(sh_degree, shs_view, d) + 0.5, 0.0)


def rasterize_dense(means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                    cov3D_precomp=None, *, image_height, image_width, tanfovx, tanfovy, bg, scale_modifier=1.0,
                    viewmatrix, projmatrix, sh_degree=0, campos, antialiasing=False, regularised_conic=True):
    """Differentiable dense forward.  Returns (color[3,H,W], radii[P], invdepth[1,H,W])."""
    dt = means3D.dtype
    P = means3D.shape[0]
    H, W = int(image_height), int(image_width)
    V = viewmatrix.to(dt)     # transposed layout: V[c, r] = math (r, c)
    Mx = projmatrix.to(dt)
    ones = torch.ones(P, 1, dtype=dt)
    ph = torch.cat([means3D, ones], dim=1)
    pview = ph @ V            # row-vector convention == the reference's (utils/graphics_utils.py:22-29)
    phom = ph @ Mx
    pw = 1.0 / (phom[:, 3] + 1e-7)
    ndc = phom[:, :2] * pw[:, None]
    if means2D is not None:
        ndc = ndc + means2D[:, :2]
    tz = pview[:, 2]
    visible = (tz > 0.2).detach()

    cov6 = cov3D_precomp if cov3D_precomp is not None else cov3d_python(scales, scale_modifier, rotations)
    S = torch.stack([cov6[:, 0], cov6[:, 1], cov6[:, 2], cov6[:, 1], cov6[:, 3], cov6[:, 4],
                     cov6[:, 2], cov6[:, 4], cov6[:, 5]], dim=-1).reshape(P, 3, 3)

    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tz_safe = torch.where(visible, tz, torch.ones_like(tz))
    txtz, tytz = pview[:, 0] / tz_safe, pview[:, 1] / tz_safe
    inx = ((txtz >= -limx) & (txtz <= limx)).detach()
    iny = ((tytz >= -limy) & (tytz <= limy)).detach()
    tx = torch.where(inx, pview[:, 0], (txtz.clamp(-limx, limx) * tz_safe).detach())
    ty = torch.where(iny, pview[:, 1], (tytz.clamp(-limy, limy) * tz_safe).detach())
    fx, fy = W / (2 * tanfovx), H / (2 * tanfovy)
    zero = torch.zeros_like(tz_safe)
    J = torch.stack([fx / tz_safe, zero, -(fx * tx) / (tz_safe * tz_safe),
                     zero, fy / tz_safe, -(fy * ty) / (tz_safe * tz_safe)], dim=-1).reshape(P, 2, 3)
    Wrot = V[:3, :3].transpose(0, 1)      # math rotation rows
    Tm = J @ Wrot                          # [P,2,3]
    cov2 = Tm @ S @ Tm.transpose(1, 2)     # [P,2,2]
    a0, b, c0 = cov2[:, 0, 0], cov2[:, 0, 1], cov2[:, 1, 1]
    det0 = a0 * c0 - b * b
    a, c = a0 + 0.3, c0 + 0.3
    det = a * c - b * b
    hconv = torch.ones_like(det)
    if antialiasing:
        hconv = torch.sqrt(torch.clamp_min(det0 / det, 0.000025))
    visible = visible & (det != 0).detach()
    det_safe = torch.where(visible, det, torch.ones_like(det))
    if regularised_conic:
        one, nil = torch.ones_like(det), torch.zeros_like(det)
        cA, cB, cC = _RegularisedConic.apply(torch.where(visible, a, one), torch.where(visible, b, nil), torch.where(visible, c, one))
    else:
        cA, cB, cC = c / det_safe, -b / det_safe, a / det_safe
    with torch.no_grad():
        mid = 0.5 * (a + c)
        lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
        lam2 = mid - torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
        radius = torch.ceil(3.0 * torch.sqrt(torch.maximum(lam, lam2)))
    pix = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    piy = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    gx, gy = (W + 15) // 16, (H + 15) // 16
    with torch.no_grad():
        def trunc_clamp(v, hi):
            return torch.clamp(torch.trunc(v), 0, hi).to(torch.int64)
        minx = trunc_clamp((pix - radius) / 16, gx); maxx = trunc_clamp((pix + radius + 15) / 16, gx)
        miny = trunc_clamp((piy - radius) / 16, gy); maxy = trunc_clamp((piy + radius + 15) / 16, gy)
        visible = visible & (((maxx - minx) * (maxy - miny)) > 0)
        radii = torch.where(visible, radius, torch.zeros_like(radius)).to(torch.int32)

    if colors_precomp is not None:
        rgb = colors_precomp
    else:
        rgb = sh_colors_python(sh_degree, shs, means3D, campos.to(dt))
    op = opacities.reshape(-1) * hconv

    # depth order, ties by index (stable sort)
    with torch.no_grad():
        depth_key = torch.where(visible, tz, torch.full_like(tz, float("inf")))
        order = torch.sort(depth_key, stable=True).indices
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    pxf, pyf = xs.reshape(-1), ys.reshape(-1)                       # [HW]
    tix, tiy = (pxf // 16).to(torch.int64), (pyf // 16).to(torch.int64)
    o = order
    dx = pix[o][None, :] - pxf[:, None]                             # [HW,P]
    dy = piy[o][None, :] - pyf[:, None]
    power = -0.5 * (cA[o][None, :] * dx * dx + cC[o][None, :] * dy * dy) - cB[o][None, :] * dx * dy
    with torch.no_grad():
        in_rect = ((tix[:, None] >= minx[o][None, :]) & (tix[:, None] < maxx[o][None, :])
                   & (tiy[:, None] >= miny[o][None, :]) & (tiy[:, None] < maxy[o][None, :]) & visible[o][None, :])
    G = torch.exp(torch.where(in_rect & (power <= 0).detach(), power, torch.zeros_like(power)))
    araw = op[o][None, :] * G
    alpha = araw + (torch.clamp_max(araw, 0.99) - araw).detach()    # straight-through (A.5 i)
    with torch.no_grad():
        valid = in_rect & (power <= 0) & (alpha >= 1.0 / 255.0)
    alpha_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
    one_minus = 1.0 - alpha_eff
    T_incl = torch.cumprod(one_minus, dim=1)
    T_excl = torch.cat([torch.ones(T_incl.shape[0], 1, dtype=dt), T_incl[:, :-1]], dim=1)
    with torch.no_grad():
        stopped = torch.cumsum((T_incl < 1e-4).to(torch.int64), dim=1) > 0
        active = valid & ~stopped
    wgt = torch.where(active, alpha_eff * T_excl, torch.zeros_like(alpha_eff))   # [HW,P]
    # final transmittance = product over applied Gaussians only
    T_final = torch.prod(torch.where(active, one_minus, torch.ones_like(one_minus)), dim=1)
    color = wgt @ rgb[o] + T_final[:, None] * bg.to(dt)[None, :]
    tz_o = torch.where(visible, tz, torch.ones_like(tz))[o]
    invdepth = wgt @ (1.0 / tz_o)
    return color.t().reshape(3, H, W), radii, invdepth.reshape(1, H, W)
