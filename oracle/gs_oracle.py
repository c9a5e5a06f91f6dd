"""ctypes front-end of oracle/gs_oracle.c (the CPU checker).

TEST INFRASTRUCTURE ONLY: importable from tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke().  The product package never imports this module.

The call surface mirrors the reference's rasterizer boundary
(renderer/gaussian_renderer/__init__.py:43-57,94-102): `rasterize()` takes the same
tensors the reference hands to `GaussianRasterizer.forward` plus the 13 settings, and
returns (color[3,H,W], radii[P], invdepth[1,H,W]) and an opaque state for `backward()`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


_VARIANTS = ("f32", "f64", "f32acc", "f32up", "f32fma")


def has_fma() -> bool:
    """The f32fma build (backward contracted into FMA3 instructions) only runs on an x86 host that has them."""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return " fma " in line + " "
    except OSError:
        pass
    return False


def build(force: bool = False) -> None:
    """Compile the C oracle (all variants of oracle/Makefile) with gcc."""
    out = os.path.join(_HERE, "_build")
    need = force or not all(os.path.exists(os.path.join(out, f"libgs_oracle_{p}.so")) for p in _VARIANTS)
    src = os.path.join(_HERE, "gs_oracle.c")
    if not need:
        newest = min(os.path.getmtime(os.path.join(out, f"libgs_oracle_{p}.so")) for p in _VARIANTS)
        need = os.path.getmtime(src) > newest
    if need:
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)


class _Scene(C.Structure):
    pass


def _scene_struct(real):
    class Scene(C.Structure):
        _fields_ = [
            ("P", C.c_int), ("D", C.c_int), ("M", C.c_int), ("W", C.c_int), ("H", C.c_int),
            ("means3D", C.c_void_p), ("shs", C.c_void_p), ("colors_precomp", C.c_void_p),
            ("opacities", C.c_void_p), ("scales", C.c_void_p), ("rotations", C.c_void_p),
            ("cov3D_precomp", C.c_void_p), ("scale_modifier", real),
            ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p),
            ("bg", C.c_void_p), ("tanfovx", real), ("tanfovy", real),
            ("antialiasing", C.c_int), ("nthreads", C.c_int),
        ]
    return Scene


def _lib(precision: str):
    if precision not in _LIBS:
        if precision == "f32fma" and not has_fma():
            raise RuntimeError("oracle variant f32fma needs a host CPU with FMA3")
        build()
        lib = C.CDLL(os.path.join(_HERE, "_build", f"libgs_oracle_{precision}.so"))
        real = C.c_double if precision == "f64" else C.c_float
        lib.or_forward.restype = C.c_void_p
        lib.or_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.or_backward.restype = None
        lib.or_backward.argtypes = [C.c_void_p] * 14
        lib.or_free.argtypes = [C.c_void_p]
        lib.or_state_N.restype = C.c_long
        lib.or_state_N.argtypes = [C.c_void_p]
        lib.or_state_interactions.restype = C.c_double
        lib.or_state_interactions.argtypes = [C.c_void_p]
        lib.or_state_copy.argtypes = [C.c_void_p] * 14
        lib.or_max_threads.restype = C.c_int
        lib.or_set_amb_policy.argtypes = [C.c_int]
        assert lib.or_real_size() == C.sizeof(real)
        _LIBS[precision] = (lib, real, _scene_struct(real))
    return _LIBS[precision]


def max_threads() -> int:
    return int(_lib("f32")[0].or_max_threads())


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@dataclass
class OracleOutput:
    color: np.ndarray       # [3,H,W]
    radii: np.ndarray       # [P] int32
    invdepth: np.ndarray    # [1,H,W]
    N: int
    interactions: float
    state: "OracleState"


class OracleState:
    """Owns the C-side state between forward and backward."""

    def __init__(self, lib, handle, scene, keep, dtype, P, W, H):
        self._lib, self._h, self._scene, self._keep = lib, handle, scene, keep
        self.dtype, self.P, self.W, self.H = dtype, P, W, H

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.or_free(self._h)
            self._h = None

    def details(self) -> dict:
        """Intermediate per-Gaussian / per-pixel buffers (for stage-wise parity tests)."""
        P, W, H = self.P, self.W, self.H
        gx, gy = (W + 15) // 16, (H + 15) // 16
        N = int(self._lib.or_state_N(self._h))
        d = dict(
            depth=np.zeros(P, self.dtype), xy=np.zeros((P, 2), self.dtype),
            conic_op=np.zeros((P, 4), self.dtype), rgb=np.zeros((P, 3), self.dtype),
            cov3D=np.zeros((P, 6), self.dtype), clamped=np.zeros((P, 3), np.uint8),
            rect=np.zeros((P, 4), np.int32), final_T=np.zeros((H, W), self.dtype),
            n_contrib=np.zeros((H, W), np.int32), gauss_ambig=np.zeros(P, np.uint8),
            pix_ambig=np.zeros((H, W), np.uint8), point_list=np.zeros(max(N, 1), np.uint32),
            ranges=np.zeros((gx * gy, 2), np.int64),
        )
        order = ["depth", "xy", "conic_op", "rgb", "cov3D", "clamped", "rect", "final_T",
                 "n_contrib", "gauss_ambig", "pix_ambig", "point_list", "ranges"]
        self._lib.or_state_copy(self._h, *[_ptr(d[k]) for k in order])
        d["point_list"] = d["point_list"][:N]
        d["N"] = N
        return d


def rasterize(*, means3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
              cov3D_precomp=None, image_height, image_width, tanfovx, tanfovy, bg, scale_modifier=1.0,
              viewmatrix, projmatrix, sh_degree=0, campos, prefiltered=False, debug=False,
              antialiasing=False, precision="f32", nthreads=0, amb_policy=0) -> OracleOutput:
    """Forward pass.  Array-likes are converted to contiguous numpy of the chosen precision."""
    lib, real, Scene = _lib(precision)
    dt = np.float64 if precision == "f64" else np.float32       # "f32acc": float32 with float32 gradient accumulators

    def arr(a):
        if a is None:
            return None
        if hasattr(a, "detach"):
            a = a.detach().cpu().numpy()
        return np.ascontiguousarray(np.asarray(a), dtype=dt)

    if (shs is None) == (colors_precomp is None):
        raise Exception("Please provide excatly one of either SHs or precomputed colors!")
    if ((scales is None or rotations is None) and cov3D_precomp is None) or \
            ((scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")

    means3D = arr(means3D); P = means3D.shape[0]
    opacities = arr(opacities).reshape(-1)
    shs = arr(shs); colors_precomp = arr(colors_precomp)
    scales = arr(scales); rotations = arr(rotations); cov3D_precomp = arr(cov3D_precomp)
    vm, pm, cp, bgc = arr(viewmatrix).reshape(-1), arr(projmatrix).reshape(-1), arr(campos).reshape(-1), arr(bg).reshape(-1)
    M = shs.shape[1] if shs is not None else 0
    W, H = int(image_width), int(image_height)
    sc = Scene(P=P, D=int(sh_degree), M=M, W=W, H=H, means3D=_ptr(means3D), shs=_ptr(shs),
               colors_precomp=_ptr(colors_precomp), opacities=_ptr(opacities), scales=_ptr(scales),
               rotations=_ptr(rotations), cov3D_precomp=_ptr(cov3D_precomp),
               scale_modifier=float(scale_modifier), viewmatrix=_ptr(vm), projmatrix=_ptr(pm),
               campos=_ptr(cp), bg=_ptr(bgc), tanfovx=float(tanfovx), tanfovy=float(tanfovy),
               antialiasing=int(bool(antialiasing)), nthreads=int(nthreads))
    keep = [means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, vm, pm, cp, bgc]
    color = np.zeros((3, H, W), dt); invd = np.zeros((1, H, W), dt); radii = np.zeros(max(P, 1), np.int32)
    # amb_policy: how pairs whose skip test is within exp() rounding are decided (0 as computed, +1 all in, -1 all out);
    # the state remembers it for the backward pass
    lib.or_set_amb_policy(int(amb_policy))
    try:
        h = lib.or_forward(C.byref(sc), _ptr(color), _ptr(invd), _ptr(radii))
    finally:
        lib.or_set_amb_policy(0)
    st = OracleState(lib, h, sc, keep, dt, P, W, H)
    return OracleOutput(color, radii[:P], invd, int(lib.or_state_N(h)), float(lib.or_state_interactions(h)), st)


def backward(out: OracleOutput, grad_color, grad_invdepth=None) -> dict:
    """Backward pass: returns the 8 gradients of the reference's autograd contract (+ dL_dconic)."""
    st = out.state
    lib, sc, dt, P = st._lib, st._scene, st.dtype, st.P

    def arr(a):
        if a is None:
            return None
        if hasattr(a, "detach"):
            a = a.detach().cpu().numpy()
        return np.ascontiguousarray(np.asarray(a), dtype=dt)

    gc = arr(grad_color); gd = arr(grad_invdepth)
    M = sc.M
    Pn = max(P, 1)
    g = dict(
        means3D=np.zeros((Pn, 3), dt), means2D=np.zeros((Pn, 3), dt), sh=np.zeros((Pn, max(M, 1), 3), dt),
        colors_precomp=np.zeros((Pn, 3), dt), opacities=np.zeros((Pn, 1), dt), scales=np.zeros((Pn, 3), dt),
        rotations=np.zeros((Pn, 4), dt), cov3D_precomp=np.zeros((Pn, 6), dt), conic=np.zeros((Pn, 4), dt),
        sh_factor=np.zeros((Pn, 3), dt),       # SH path: clamp-masked dL/dcolour (dL/dsh[k][c] = Y_k(dir) * sh_factor[c])
    )
    lib.or_backward(C.byref(sc), st._h, _ptr(gc), _ptr(gd), _ptr(g["means3D"]), _ptr(g["means2D"]),
                    _ptr(g["sh"]) if M > 0 else None, _ptr(g["colors_precomp"]), _ptr(g["opacities"]),
                    _ptr(g["scales"]), _ptr(g["rotations"]), _ptr(g["cov3D_precomp"]), _ptr(g["conic"]), _ptr(g["sh_factor"]))
    return {k: v[:P] for k, v in g.items()}
