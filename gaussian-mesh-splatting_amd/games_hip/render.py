"""`render()` with the signature and return dict of renderer/gaussian_renderer/__init__.py:25-111.

On a machine that has the reference checked out, the reference's own renderer modules run unchanged
on top of the drop-in `diff_gaussian_rasterization`; this restatement exists because bench.py and the
GPU tests run where the reference tree is absent.  Flag semantics (`compute_cov3D_python`,
`convert_SHs_python`, `debug`, `antialiasing`) follow arguments/__init__.py:64-70.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer


@dataclass
class PipelineParams:
    convert_SHs_python: bool = False
    compute_cov3D_python: bool = False
    debug: bool = False
    antialiasing: bool = False


# python stages of the reference, restated device-agnostically (scene/gaussian_model.py:27-31,
# utils/general_utils.py:144-190, utils/sh_utils.py:57-112): the drop-in must give the same image
# whichever side computes them, which tests/test_gpu_raster.py checks.
def covariance_python(scaling, scaling_modifier, rotation):
    q = rotation / rotation.norm(dim=1, keepdim=True)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(-1, 3, 3)
    L = R * (scaling_modifier * scaling)[:, None, :]
    S = L @ L.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=-1)


_C0, _C1 = 0.28209479177387814, 0.4886025119029199
_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
       1.445305721320277, -0.5900435899266435]


def eval_sh(deg, sh, dirs):
    result = _C0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        result = result - _C1 * y * sh[..., 1] + _C1 * z * sh[..., 2] - _C1 * x * sh[..., 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            result = (result + _C2[0] * xy * sh[..., 4] + _C2[1] * yz * sh[..., 5] + _C2[2] * (2.0 * zz - xx - yy) * sh[..., 6]
                      + _C2[3] * xz * sh[..., 7] + _C2[4] * (xx - yy) * sh[..., 8])
            if deg > 2:
                result = (result + _C3[0] * y * (3 * xx - yy) * sh[..., 9] + _C3[1] * xy * z * sh[..., 10]
                          + _C3[2] * y * (4 * zz - xx - yy) * sh[..., 11] + _C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12]
                          + _C3[4] * x * (4 * zz - xx - yy) * sh[..., 13] + _C3[5] * z * (xx - yy) * sh[..., 14]
                          + _C3[6] * x * (xx - 3 * yy) * sh[..., 15])
    return result


_zero_cache = {}


def _zero_points(xyz: torch.Tensor) -> torch.Tensor:
    key = (xyz.device, tuple(xyz.shape), xyz.dtype)
    z = _zero_cache.get(key)
    if z is None:
        if len(_zero_cache) >= 8:
            _zero_cache.clear()
        z = _zero_cache[key] = torch.zeros(xyz.shape, dtype=xyz.dtype, device=xyz.device)
    return z.detach().requires_grad_(True)


def _fused_training_ok(pc, pipe, override_color) -> bool:
    """Can this DIFFERENTIATED frame be rendered straight from the mesh (games_hip.model.HipMeshMixin.hip_defer_k0)?  The model
    deferred its K0 (update_alpha() since the last optimizer step, nothing has asked for the derived values), uniform splats per
    face, split degree-3 SH STORAGE (any active degree), the rasterizer's native SH / cov3D stages."""
    import os
    import diff_gaussian_rasterization as dgr
    if not (hasattr(pc, "__dict__") and pc.__dict__.get("_hip_pending")) or not torch.is_grad_enabled():
        return False
    if dgr._C is None or not hasattr(dgr._C, "render_mesh") or os.environ.get("GMS_TRAIN_FUSED", "1") == "0":
        return False
    if override_color is not None or pipe.compute_cov3D_python or pipe.convert_SHs_python:
        return False
    a, fr, dc, op = getattr(pc, "_alpha", None), getattr(pc, "_features_rest", None), getattr(pc, "_features_dc", None), getattr(pc, "_opacity", None)
    if not (torch.is_tensor(a) and a.dim() == 3 and a.is_cuda and torch.is_tensor(fr) and fr.dim() == 3 and fr.shape[1] == 15 and torch.is_tensor(dc)):
        return False
    P = int(a.shape[0] * a.shape[1])
    sc = getattr(pc, getattr(pc, "_hip_scale_attr", "_scale"), None)
    return (0 <= int(pc.active_sh_degree) <= 3 and torch.is_tensor(op) and op.numel() == P and torch.is_tensor(sc) and sc.numel() == P
            and dc.is_contiguous() and fr.is_contiguous() and dc.shape[0] == P and torch.is_tensor(getattr(pc, "faces", None)))


def _render_training_frame_from_mesh(viewpoint_camera, pc, pipe, bg_color, scaling_modifier):
    """train.py:100 on a model whose K0 is deferred: ONE C++ autograd node from (vertices, _alpha, _scale, _opacity, SH) to the image
    (`_C.render_mesh`; GmsRasterForwardArgs.mesh + mesh_out_*).  Same image and gradients as the two-node graph (tests/test_gpu_fused_training.py)."""
    import diff_gaussian_rasterization as dgr
    from .mesh_op import ALPHA_MODES
    vertices, faces, _alpha, _scale = pc._hip_inputs()
    if faces.dtype != torch.int64 or not faces.is_contiguous():
        faces = faces.long().contiguous()
    P = int(_alpha.shape[0] * _alpha.shape[1])
    key = (_alpha.device, (P, 3), torch.float32)
    z = _zero_cache.get(key)
    if z is None:
        if len(_zero_cache) >= 8:
            _zero_cache.clear()
        z = _zero_cache[key] = torch.zeros((P, 3), dtype=torch.float32, device=_alpha.device)
    screenspace_points = z.detach().requires_grad_(True)
    H, W = int(viewpoint_camera.image_height), int(viewpoint_camera.image_width)
    image, radii, invdepth, xyz, scaling_act, rotation_unit, opacity_act, visible = dgr._C.render_mesh(
        vertices, faces, _alpha, _scale, pc._opacity, pc._features_dc, pc._features_rest, screenspace_points,
        ALPHA_MODES[getattr(pc, "alpha_mode", "relu")], int(_alpha.shape[1]), dgr._empty(_alpha.device), bg_color,
        viewpoint_camera.world_view_transform, viewpoint_camera.full_proj_transform, viewpoint_camera.camera_center, H, W,
        math.tan(viewpoint_camera.FoVx * 0.5), math.tan(viewpoint_camera.FoVy * 0.5), float(scaling_modifier), bool(pipe.antialiasing),
        bool(pipe.debug), int(pc.active_sh_degree))          # (the ACTIVE degree: train.py:86-87 raises it every 1 000 iterations)
    pc._hip_fused_frame(xyz, scaling_act, rotation_unit, opacity_act)
    # (`visibility_filter` = radii > 0 comes out of the preprocess kernel, as on the two-node route: no elementwise launch)
    return {"render": image, "viewspace_points": screenspace_points, "visibility_filter": visible, "radii": radii, "depth": invdepth}


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None):
    if _fused_training_ok(pc, pipe, override_color):
        return _render_training_frame_from_mesh(viewpoint_camera, pc, pipe, bg_color, scaling_modifier)
    xyz = pc.get_xyz
    # the reference writes `torch.zeros_like(...) + 0` and retain_grad() (renderer/gaussian_renderer/__init__.py:33-37): a
    # 3.6 MB fill + an elementwise kernel per render for a tensor whose VALUES nobody reads (the rasterizer only hands a
    # gradient back through it).  Here: a fresh leaf aliasing a cached all-zero buffer -- same values, its own `.grad`
    # (train.py:129-133 reads it for the densification statistics), no kernel
    screenspace_points = _zero_points(xyz)
    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center, prefiltered=False,
        debug=pipe.debug, antialiasing=pipe.antialiasing)
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)
    means3D, means2D, opacity = xyz, screenspace_points, pc.get_opacity
    scales = rotations = cov3D_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = covariance_python(pc.get_scaling, scaling_modifier, pc._rotation)
    else:
        scales, rotations = pc.get_scaling, pc.get_rotation
    shs = colors_precomp = None
    if override_color is None:
        if pipe.convert_SHs_python:
            shs_view = pc.get_features.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
            dir_pp = xyz - viewpoint_camera.camera_center.repeat(pc.get_features.shape[0], 1)
            dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
            colors_precomp = torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dir_pp_normalized) + 0.5, 0.0)
        else:
            shs = pc.get_features
    else:
        colors_precomp = override_color
    rendered_image, radii, depth_image = rasterizer(
        means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp, opacities=opacity,
        scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    visible = getattr(rasterizer, "visibility_filter", None)      # written by the preprocess kernel (this drop-in only)
    return {"render": rendered_image, "viewspace_points": screenspace_points,
            "visibility_filter": visible if visible is not None else radii > 0,
            "radii": radii, "depth": depth_image}


def _fused_frame_ok(pc, pipe, override_color) -> bool:
    """Can this forward-only frame take the fused mesh -> image path (GmsRasterForwardArgs.mesh)?  Uniform splats per face, split
    degree-3 SH storage at full degree, the rasterizer's native SH / cov3D stages, nothing to differentiate."""
    import os
    import diff_gaussian_rasterization as dgr
    if dgr._C is None or not hasattr(dgr._C, "render_mesh_forward") or os.environ.get("GMS_ANIMATE_FUSED", "1") == "0":
        return False
    if torch.is_grad_enabled() or override_color is not None or pipe.compute_cov3D_python or pipe.convert_SHs_python:
        return False
    a, fr, op = getattr(pc, "_alpha", None), getattr(pc, "_features_rest", None), getattr(pc, "_opacity", None)
    if not (torch.is_tensor(a) and a.dim() == 3 and a.is_cuda and torch.is_tensor(fr) and fr.dim() == 3 and fr.shape[1] == 15):
        return False
    P = int(a.shape[0] * a.shape[1])
    sc = getattr(pc, getattr(pc, "_hip_scale_attr", "_scale"), None)
    # the unfused frame reads get_opacity: the kernel's sigmoid while the model's cache of it is current (else torch.sigmoid, which
    # this path does not reproduce bit for bit)
    cached = pc.__dict__.get("_hip_opacity") if hasattr(pc, "__dict__") else None
    cache_ok = cached is not None and cached[0] is op and cached[1] == op._version
    return (int(pc.active_sh_degree) == 3 and torch.is_tensor(op) and op.numel() == P and torch.is_tensor(sc) and sc.numel() == P
            and cache_ok and pc._features_dc.is_contiguous() and fr.is_contiguous())


def render_mesh_frame(vertices: torch.Tensor, faces: torch.Tensor, viewpoint_camera, pc, pipe, bg_color: torch.Tensor,
                      scaling_modifier=1.0):
    """One forward-only frame straight from a (deformed) mesh: vertices [V,3] + faces [F,3] -> image, with the face -> Gaussian
    parameterization computed inside the rasterizer's preprocess thread (no K0 launch, no xyz / scale / rotation tensors; SURVEY.md
    section 7 step 9).  Same image, bit for bit, as `render_animated(None, vertices[faces], ...)`; callers check `_fused_frame_ok`.
    Side effects differ from the unfused route ON PURPOSE (the frame materialises no per-Gaussian tensor): `pc._scaling` / `pc._rotation`
    / `pc._xyz` keep the values of the last `prepare_scaling_rot()` (the undeformed mesh), and `viewspace_points` is None -- a
    forward-only frame has no screen-space gradient to receive.  Code that saves or inspects the DEFORMED Gaussians after a frame
    calls `render_animated` with `GMS_ANIMATE_FUSED=0` (or assigns `pc.triangles` and calls `prepare_scaling_rot()`)."""
    import diff_gaussian_rasterization as dgr
    from .mesh_op import ALPHA_MODES
    H, W = int(viewpoint_camera.image_height), int(viewpoint_camera.image_width)
    color, radii, invdepth, visible = dgr._C.render_mesh_forward(
        vertices.float(), faces, pc._alpha, getattr(pc, getattr(pc, "_hip_scale_attr", "_scale")), pc._opacity, ALPHA_MODES[getattr(pc, "alpha_mode", "relu")], int(pc._alpha.shape[1]),
        dgr._empty(vertices.device), pc._features_dc, pc._features_rest, bg_color, viewpoint_camera.world_view_transform,
        viewpoint_camera.full_proj_transform, viewpoint_camera.camera_center, H, W, math.tan(viewpoint_camera.FoVx * 0.5),
        math.tan(viewpoint_camera.FoVy * 0.5), float(scaling_modifier), bool(pipe.antialiasing), bool(pipe.debug))
    return {"render": color, "viewspace_points": None, "visibility_filter": visible, "radii": radii, "depth": invdepth}


def render_animated(idxs, triangles, viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0,
                    override_color=None):
    """renderer/gaussian_animated_renderer/__init__.py:21-121: the mesh is deformed per frame, centres follow
    `pc.alpha @ triangles` (:61-67) and scale / rotation are re-derived from the deformed triangles (:72-73).
    With the fused op that is one kernel: assigning `pc.triangles` and calling `prepare_scaling_rot()` runs
    the op on the explicit triangles; its xyz output is `alpha @ triangles`."""
    from .mesh_op import triangles_to_gaussians, _identity_faces
    if _fused_frame_ok(pc, pipe, override_color):
        # forward-only frame (the animated drivers run under no_grad): K0 inside the preprocess thread, explicit triangles as an
        # identity-indexed mesh.  (The cached kernel sigmoid is what the unfused frame would read from get_opacity.)
        F_ = int(triangles.shape[0])
        if F_ != int(pc._alpha.shape[0]):          # (the unfused route fails in `alpha @ triangles`: same error class here)
            raise RuntimeError(f"render_animated: {F_} triangles for a model with {int(pc._alpha.shape[0])} faces")
        key = (triangles.device, F_)
        if key not in _identity_faces:
            _identity_faces[key] = torch.arange(3 * F_, device=triangles.device, dtype=torch.int64).reshape(F_, 3)
        pc.triangles = triangles
        return render_mesh_frame(triangles.reshape(3 * F_, 3), _identity_faces[key], viewpoint_camera, pc, pipe, bg_color, scaling_modifier)
    pc.triangles = triangles
    _, xyz, scaling, rotation, scaling_act, rotation_unit = triangles_to_gaussians(
        triangles, pc._alpha, pc._scale, getattr(pc, "alpha_mode", "relu"), fused_activations=True)
    pc._scaling, pc._rotation = scaling, rotation
    pc._hip_activated = (scaling, rotation, scaling_act, rotation_unit)

    class _View:       # same model, centres from the deformed mesh (the reference passes them as means3D)
        def __getattr__(self, name):
            return getattr(pc, name)
        get_xyz = xyz
    return render(viewpoint_camera, _View(), pipe, bg_color, scaling_modifier, override_color)
