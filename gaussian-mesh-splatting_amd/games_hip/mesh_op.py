"""Fused mesh-face -> Gaussian parameterization op (K0) on top of libgmsplat.so.

Replaces, with identical results, the python of
  GaussianMeshModel.update_alpha / _calc_xyz / prepare_scaling_rot
      games/mesh_splatting/scene/gaussian_mesh_model.py:86-169
  rot_to_quat_batch                                  utils/general_utils.py:43-96
  the per-mesh loop of GaussianMultiMeshModel        games/multi_mesh_splatting/scene/gaussian_multi_mesh_model.py:99-199
  the softmax-alpha variant of GaussianFlameModel    games/flame_splatting/scene/gaussian_flame_model.py:195
One forward kernel, two backward kernels; autograd sees a single node.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from diff_gaussian_rasterization import _lib

ALPHA_MODES = {"relu": _lib.GMS_ALPHA_RELU, "softmax": _lib.GMS_ALPHA_SOFTMAX}


def _c(t, dtype):
    if t.dtype != dtype:
        t = t.to(dtype)
    return t if t.is_contiguous() else t.contiguous()


def _mesh_args(vertices, faces, _alpha, _scale, mode, splats_per_face, face_splat_offset, splat_face, fused=False,
               _opacity=None, prezero=None, prezeroed=False):
    P = _alpha.shape[0] if _alpha.dim() == 2 else _alpha.shape[0] * _alpha.shape[1]
    return _lib.MeshArgs(F=int(faces.shape[0]), V=int(vertices.shape[0]), P=int(P), splats_per_face=int(splats_per_face),
                         alpha_mode=int(mode), vertices=_lib.ptr(vertices), faces=_lib.ptr(faces),
                         face_splat_offset=_lib.ptr(face_splat_offset), splat_face=_lib.ptr(splat_face),
                         _alpha=_lib.ptr(_alpha), _scale=_lib.ptr(_scale), fused_activations=int(bool(fused)),
                         _opacity=_lib.ptr(_opacity), prezero=_lib.ptr(prezero),
                         prezero_count=int(prezero.numel()) if prezero is not None else 0,
                         vertex_grad_prezeroed=int(bool(prezeroed)))


class _MeshToGaussians(torch.autograd.Function):
    """outputs: alpha, xyz, scaling (log), rotation, and -- when `fused` -- exp(scaling), normalize(rotation)
    [, sigmoid(_opacity) when `_opacity` is given].  With `fused` the gradient enters through the activated outputs
    (the log / raw ones are then attribute-parity outputs only and carry no gradient)."""

    @staticmethod
    def forward(ctx, vertices, faces, _alpha, _scale, mode, splats_per_face, face_splat_offset, splat_face, fused,
                _opacity=None):
        lib = _lib.load()
        _lib.require_gpu(vertices, faces, _alpha, _scale)
        device = vertices.device
        ctx.set_materialize_grads(False)            # unused outputs arrive as None, not as zero-filled tensors
        if _opacity is not None:
            if not fused:
                raise ValueError("_opacity fusion needs fused_activations=True")
            _opacity = _c(_opacity, torch.float32)
        vertices, _alpha, _scale = _c(vertices, torch.float32), _c(_alpha, torch.float32), _c(_scale, torch.float32)
        faces = _c(faces, torch.int64)
        if face_splat_offset is not None:
            face_splat_offset, splat_face = _c(face_splat_offset, torch.int32), _c(splat_face, torch.int32)
        P = _scale.numel()
        alpha = torch.empty_like(_alpha)
        xyz = torch.empty((P, 3), dtype=torch.float32, device=device)
        scaling = torch.empty((P, 3), dtype=torch.float32, device=device)
        rotation = torch.empty((P, 4), dtype=torch.float32, device=device)
        scaling_act = torch.empty((P, 3), dtype=torch.float32, device=device) if fused else None
        rotation_unit = torch.empty((P, 4), dtype=torch.float32, device=device) if fused else None
        opacity_act = torch.empty_like(_opacity) if _opacity is not None else None
        # the vertex-gradient buffer of the coming backward is cleared by spare blocks of this launch, so the backward's
        # per-splat and per-face parts can share one launch (a second backward through the same graph allocates afresh)
        ctx.vertex_grad = torch.empty_like(vertices) if ctx.needs_input_grad[0] else None
        a = _mesh_args(vertices, faces, _alpha, _scale, mode, splats_per_face, face_splat_offset, splat_face, fused, _opacity,
                       prezero=ctx.vertex_grad)
        with _lib.on_device(device):
            stream = _lib.stream_ptr(device)
            _lib.check(lib.gms_mesh_to_gaussians_forward(C.byref(a), _lib.ptr(alpha), _lib.ptr(xyz), _lib.ptr(scaling),
                                                         _lib.ptr(rotation), _lib.ptr(scaling_act), _lib.ptr(rotation_unit),
                                                         _lib.ptr(opacity_act), C.c_void_p(stream)),
                       "gms_mesh_to_gaussians_forward")
        ctx.save_for_backward(vertices, faces, _alpha, _scale,
                              face_splat_offset if face_splat_offset is not None else torch.empty(0, device=device),
                              splat_face if splat_face is not None else torch.empty(0, device=device),
                              _opacity if _opacity is not None else torch.empty(0, device=device))
        ctx.mode, ctx.spf, ctx.fused = mode, splats_per_face, fused
        if fused:
            ctx.mark_non_differentiable(alpha, scaling, rotation)
            if _opacity is not None:
                return alpha, xyz, scaling, rotation, scaling_act, rotation_unit, opacity_act
            return alpha, xyz, scaling, rotation, scaling_act, rotation_unit
        ctx.mark_non_differentiable(alpha)
        return alpha, xyz, scaling, rotation

    @staticmethod
    def backward(ctx, _g_alpha, g_xyz, g_scaling, g_rotation, g_scaling_act=None, g_rotation_unit=None, g_opacity_act=None):
        lib = _lib.load()
        vertices, faces, _alpha, _scale, fso, sf, _opacity = ctx.saved_tensors
        device = vertices.device
        P = _scale.numel()
        fso = fso if fso.numel() else None
        sf = sf if sf.numel() else None
        _opacity = _opacity if _opacity.numel() else None
        if ctx.fused:
            g_scaling, g_rotation = g_scaling_act, g_rotation_unit

        def grad_or_zero(g, shape):
            return torch.zeros(shape, dtype=torch.float32, device=device) if g is None else _c(g, torch.float32)

        g_xyz, g_scaling, g_rotation = grad_or_zero(g_xyz, (P, 3)), grad_or_zero(g_scaling, (P, 3)), grad_or_zero(g_rotation, (P, 4))
        d_vertices, prezeroed = ctx.vertex_grad, ctx.vertex_grad is not None
        ctx.vertex_grad = None                       # handed to autograd: never reused
        if d_vertices is None:
            d_vertices = torch.empty_like(vertices)  # cleared by the first backward kernel
        d_alpha = torch.empty_like(_alpha)
        d_scale = torch.empty_like(_scale)
        d_opacity = None
        if _opacity is not None and g_opacity_act is not None:
            g_opacity_act = _c(g_opacity_act, torch.float32)
            d_opacity = torch.empty_like(_opacity)
        a = _mesh_args(vertices, faces, _alpha, _scale, ctx.mode, ctx.spf, fso, sf, ctx.fused, _opacity, prezeroed=prezeroed)
        with _lib.on_device(device):
            stream = _lib.stream_ptr(device)
            _lib.check(lib.gms_mesh_to_gaussians_backward(C.byref(a), _lib.ptr(g_xyz), _lib.ptr(g_scaling), _lib.ptr(g_rotation),
                                                          _lib.ptr(g_opacity_act) if d_opacity is not None else None,
                                                          _lib.ptr(d_vertices), _lib.ptr(d_alpha), _lib.ptr(d_scale),
                                                          _lib.ptr(d_opacity), C.c_void_p(stream)),
                       "gms_mesh_to_gaussians_backward")
        return d_vertices, None, d_alpha, d_scale, None, None, None, None, None, d_opacity


def mesh_to_gaussians(vertices: torch.Tensor, faces: torch.Tensor, _alpha: torch.Tensor, _scale: torch.Tensor,
                      alpha_mode: str = "relu", face_splat_offset: Optional[torch.Tensor] = None,
                      splat_face: Optional[torch.Tensor] = None, fused_activations: bool = False,
                      _opacity: Optional[torch.Tensor] = None):
    """(alpha, _xyz[P,3], _scaling[P,3] (log), _rotation[P,4]) for mesh-bound Gaussians.

    `_alpha` is [F,S,3] (uniform S splats per face, the single-mesh models) or [P,3] together with
    CSR `face_splat_offset` [F+1] / `splat_face` [P] (concatenated meshes with different S).
    `fused_activations=True` appends (exp(_scaling), normalize(_rotation)) -- the property getters of
    scene/gaussian_model.py:95-101 -- computed in the same kernel and differentiated in the same backward; with
    `_opacity` [P,1] also sigmoid(_opacity) (`get_opacity`, scene/gaussian_model.py:113-115) as a seventh output."""
    mode = ALPHA_MODES[alpha_mode]
    if face_splat_offset is None:
        if _alpha.dim() != 3:
            raise ValueError("_alpha must be [F,S,3] when no face_splat_offset is given")
        spf = int(_alpha.shape[1])
    else:
        spf = 0
    import diff_gaussian_rasterization as _dgr
    if _dgr._C is not None:          # autograd node in C++ (csrc/torch_binding.cpp), same C ABI underneath
        if _opacity is not None and not fused_activations:
            raise ValueError("_opacity fusion needs fused_activations=True")
        e = _dgr._empty(vertices.device)
        return tuple(_dgr._C.mesh_to_gaussians(vertices, faces, _alpha, _scale, mode, spf,
                                               face_splat_offset if face_splat_offset is not None else e,
                                               splat_face if splat_face is not None else e, bool(fused_activations),
                                               _opacity if _opacity is not None else e))
    return _MeshToGaussians.apply(vertices, faces, _alpha, _scale, mode, spf, face_splat_offset, splat_face,
                                  bool(fused_activations), _opacity)


_identity_faces = {}


def triangles_to_gaussians(triangles: torch.Tensor, _alpha: torch.Tensor, _scale: torch.Tensor, alpha_mode: str = "relu",
                           fused_activations: bool = False):
    """Same op driven by explicit triangles [F,3,3] (the animated renderers replace `pc.triangles`
    per frame: renderer/gaussian_animated_renderer/__init__.py:61-73)."""
    F_ = int(triangles.shape[0])
    key = (triangles.device, F_)
    if key not in _identity_faces:
        _identity_faces[key] = torch.arange(3 * F_, device=triangles.device, dtype=torch.int64).reshape(F_, 3)
    return mesh_to_gaussians(triangles.reshape(3 * F_, 3), _identity_faces[key], _alpha, _scale, alpha_mode,
                             fused_activations=fused_activations)
