"""CPU restatement of GaMeS's mesh-face -> Gaussian parameterization (K0).

TEST INFRASTRUCTURE ONLY.  Parity PINNED: this restatement is checked against the reference's
own classes executed in the authoring container (tests/golden/make_golden.py imports
games/mesh_splatting/scene/gaussian_mesh_model.py through oracle/ref_import.py and commits
tests/golden/k0_*.npz; tests/test_mesh_oracle.py compares).

Follows, line by line:
  update_alpha        games/mesh_splatting/scene/gaussian_mesh_model.py:153-169
  _calc_xyz           games/mesh_splatting/scene/gaussian_mesh_model.py:86-101
  prepare_scaling_rot games/mesh_splatting/scene/gaussian_mesh_model.py:103-151
  rot_to_quat_batch   utils/general_utils.py:43-96 (+ _sqrt_positive_part :33-41,
                      standardize_quaternion :19-31)
  softmax alpha       games/flame_splatting/scene/gaussian_flame_model.py:195
  multi-mesh          games/multi_mesh_splatting/scene/gaussian_multi_mesh_model.py:99-199
Written with differentiable torch ops (any dtype/device) so autograd provides the reference
gradient: the reference's own backward is autograd through the same op sequence.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch

EPS_S0 = 1e-8


def _sqrt_positive_part(x):
    """sqrt(max(0,x)) with zero subgradient where x <= 0 (utils/general_utils.py:33-41)."""
    pos = x > 0
    safe = torch.where(pos, x, torch.ones_like(x))
    return torch.where(pos, torch.sqrt(safe), torch.zeros_like(x))


def rot_to_quat_batch(rot):
    m = rot.reshape(-1, 9)
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(m, dim=-1)
    q_abs = _sqrt_positive_part(torch.stack([
        1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], dim=-1))
    quat_by_rijk = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1)], dim=-2)
    flr = torch.tensor(0.1, dtype=q_abs.dtype, device=q_abs.device)
    cand = quat_by_rijk / (2.0 * q_abs[..., None].max(flr))
    sel = q_abs.argmax(dim=-1)                                  # ties -> first index
    out = torch.gather(cand, 1, sel[:, None, None].expand(-1, 1, 4)).squeeze(1)
    return torch.where(out[..., 0:1] < 0, -out, out)


def face_frames(triangles):
    """Per-face (v0,v1,v2) frame and scales (s0,s1,s2): gaussian_mesh_model.py:124-141."""
    def dot(v, u):
        return (v * u).sum(dim=-1, keepdim=True)

    normals = torch.linalg.cross(triangles[:, 1] - triangles[:, 0], triangles[:, 2] - triangles[:, 0], dim=1)
    v0 = normals / (torch.linalg.vector_norm(normals, dim=-1, keepdim=True) + EPS_S0)
    means = torch.mean(triangles, dim=1)
    v1 = triangles[:, 1] - means
    v1_norm = torch.linalg.vector_norm(v1, dim=-1, keepdim=True) + EPS_S0
    v1 = v1 / v1_norm
    v2_init = triangles[:, 2] - means
    v2 = v2_init - dot(v2_init, v0) * v0 - dot(v2_init, v1) * v1
    v2 = v2 / (torch.linalg.vector_norm(v2, dim=-1, keepdim=True) + EPS_S0)
    s1 = v1_norm / 2.0
    s2 = dot(v2_init, v2) / 2.0
    s0 = EPS_S0 * torch.ones_like(s1)
    return v0, v1, v2, torch.cat((s0, s1, s2), dim=1)


def mesh_to_gaussians(vertices, faces, _alpha, _scale, alpha_mode: str = "relu"
                      ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """Returns (alpha[F,S,3], triangles[F,3,3], _xyz[P,3], _scaling[P,3] (log), _rotation[P,4] (w,x,y,z))."""
    F_, S = _alpha.shape[0], _alpha.shape[1]
    if alpha_mode == "relu":
        alpha = torch.relu(_alpha) + 1e-8
        alpha = alpha / alpha.sum(dim=-1, keepdim=True)
    elif alpha_mode == "softmax":
        alpha = torch.softmax(_alpha, dim=2)
    else:
        raise ValueError(alpha_mode)
    triangles = vertices[faces]
    xyz = torch.matmul(alpha, triangles).reshape(F_ * S, 3)
    v0, v1, v2, scales = face_frames(triangles)
    scales = scales.unsqueeze(1).broadcast_to((F_, S, 3)).flatten(0, 1)
    scaling = torch.log(torch.relu(_scale * scales) + EPS_S0)
    rotation = torch.stack((v0, v1, v2), dim=1).unsqueeze(1)
    rotation = rotation.broadcast_to((F_, S, 3, 3)).flatten(0, 1).transpose(-2, -1)
    return alpha, triangles, xyz, scaling, rot_to_quat_batch(rotation)


def multi_mesh_to_gaussians(vertices: Sequence[torch.Tensor], faces: Sequence[torch.Tensor],
                            _alpha: Sequence[torch.Tensor], _scale: Sequence[torch.Tensor]):
    """Loop-and-concatenate exactly as gaussian_multi_mesh_model.py:99-199."""
    xs: List[torch.Tensor] = []
    ss: List[torch.Tensor] = []
    rs: List[torch.Tensor] = []
    for v, f, a, s in zip(vertices, faces, _alpha, _scale):
        _, _, x, sc, r = mesh_to_gaussians(v, f, a, s, "relu")
        xs.append(x); ss.append(sc); rs.append(r)
    return torch.cat(xs), torch.cat(ss), torch.cat(rs)


def activated(xyz, scaling, rotation, _opacity, f_dc, f_rest):
    """Property getters feeding the rasterizer: scene/gaussian_model.py:95-115."""
    return (xyz, torch.exp(scaling), torch.nn.functional.normalize(rotation), torch.sigmoid(_opacity),
            torch.cat((f_dc, f_rest), dim=1))
