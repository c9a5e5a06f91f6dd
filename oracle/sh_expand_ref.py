"""Test infrastructure: torch restatement of gms_sh_grad_expand (include/gmsplat.h) -- the SH gradient of a multi-view
step from per-view colour-gradient factors,  dL/dsh[i][k][c] = sum_v Y_k(normalize(x_i - campos_v)) * factor_v[i][c].

The basis follows the reference's utils/sh_utils.py:57-112 (`eval_sh`: C0, C1, C2[5], C3[7] and the polynomial of each term),
which is also what the rasterizer's SH evaluation uses.  Only tests import this module."""
import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658, 1.445305721320277,
      -0.5900435899266435]


def sh_basis(deg: int, dirs: torch.Tensor) -> torch.Tensor:
    """[P,(deg+1)^2] basis values for unit directions [P,3]."""
    x, y, z = dirs[:, 0], dirs[:, 1], dirs[:, 2]
    Y = [torch.full_like(x, C0)]
    if deg > 0:
        Y += [-C1 * y, C1 * z, -C1 * x]
    if deg > 1:
        xx, yy, zz = x * x, y * y, z * z
        Y += [C2[0] * x * y, C2[1] * y * z, C2[2] * (2.0 * zz - xx - yy), C2[3] * x * z, C2[4] * (xx - yy)]
    if deg > 2:
        Y += [C3[0] * y * (3 * xx - yy), C3[1] * x * y * z, C3[2] * y * (4 * zz - xx - yy), C3[3] * z * (2 * zz - 3 * xx - 3 * yy),
              C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy), C3[6] * x * (xx - 3 * yy)]
    return torch.stack(Y, dim=1)


def expand(factors: torch.Tensor, means3D: torch.Tensor, deg: int, M: int) -> torch.Tensor:
    """factors [V,P+1,3] (row P of each view = its camera centre) -> dL/dsh [P,M,3]; coefficients above `deg` stay zero."""
    V, P1, _ = factors.shape
    P = P1 - 1
    out = torch.zeros((P, M, 3), dtype=factors.dtype)
    for v in range(V):
        g, cam = factors[v, :P], factors[v, P]
        d = means3D.to(factors.dtype) - cam[None]
        d = d / d.norm(dim=1, keepdim=True)
        Y = sh_basis(deg, d)                                   # [P,nb]
        out[:, :Y.shape[1]] += Y[:, :, None] * g[:, None, :]
    return out
