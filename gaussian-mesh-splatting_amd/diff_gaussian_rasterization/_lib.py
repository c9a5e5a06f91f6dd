"""ctypes binding of libgmsplat.so (the C ABI declared in include/gmsplat.h).

The product path has NO CPU fallback: if the HIP library is missing or the tensors are not on a
GPU, calls fail loudly.  Build with `make -C gaussian-mesh-splatting_amd/csrc` (or
`python __graft_entry__.py build`).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GMSPLAT_LIB", os.path.join(os.path.dirname(_HERE), "lib", "libgmsplat.so"))

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)

GMS_ABI_VERSION = 8
GMS_ALPHA_RELU, GMS_ALPHA_SOFTMAX = 0, 1
ERRORS = {-1: "invalid argument", -2: "scratch allocation failed", -3: "HIP runtime error", -4: "capacity"}


class RasterForwardArgs(C.Structure):
    _fields_ = [
        ("P", C.c_int32), ("D", C.c_int32), ("M", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
        ("background", C.c_void_p), ("means3D", C.c_void_p), ("shs", C.c_void_p), ("shs_rest", C.c_void_p),
        ("colors_precomp", C.c_void_p),
        ("opacities", C.c_void_p), ("scales", C.c_void_p), ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p),
        ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p),
        ("scale_modifier", C.c_float), ("tan_fovx", C.c_float), ("tan_fovy", C.c_float),
        ("prefiltered", C.c_int32), ("antialiasing", C.c_int32), ("debug", C.c_int32),
        ("out_color", C.c_void_p), ("out_invdepth", C.c_void_p), ("radii", C.c_void_p),
        ("geom_alloc", ALLOC_FN), ("geom_ctx", C.c_void_p),
        ("binning_alloc", ALLOC_FN), ("binning_ctx", C.c_void_p),
        ("image_alloc", ALLOC_FN), ("image_ctx", C.c_void_p),
        ("binning_capacity_hint", C.c_int64), ("visible", C.c_void_p), ("num_units_out", C.c_void_p), ("no_host_wait", C.c_int32),
        ("mesh", C.c_void_p),          # ABI 5: const GmsMeshArgs * (frame straight from a mesh) or NULL
        # ABI 6: what the fused frame derived, stored for the backward (all NULL: a forward-only frame)
        ("mesh_out_xyz", C.c_void_p), ("mesh_out_scaling_act", C.c_void_p), ("mesh_out_rotation_unit", C.c_void_p), ("mesh_out_opacity_act", C.c_void_p),
        ("count_ticket_out", C.c_void_p),          # ABI 7: host int64*, deferred read-back of the frame's counts (gms_rasterize_forward_counts)
    ]


class RasterBackwardArgs(C.Structure):
    _fields_ = [
        ("P", C.c_int32), ("D", C.c_int32), ("M", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
        ("num_rendered", C.c_int64), ("binning_capacity", C.c_int64),
        ("background", C.c_void_p),
        ("means3D", C.c_void_p), ("shs", C.c_void_p), ("shs_rest", C.c_void_p), ("colors_precomp", C.c_void_p),
        ("opacities", C.c_void_p),
        ("scales", C.c_void_p), ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p),
        ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p),
        ("scale_modifier", C.c_float), ("tan_fovx", C.c_float), ("tan_fovy", C.c_float),
        ("antialiasing", C.c_int32), ("debug", C.c_int32),
        ("radii", C.c_void_p), ("geom_buffer", C.c_void_p), ("binning_buffer", C.c_void_p), ("image_buffer", C.c_void_p),
        ("dL_dout_color", C.c_void_p), ("dL_dout_invdepth", C.c_void_p),
        ("grad_accum", C.c_void_p),
        ("dL_dmeans2D", C.c_void_p), ("dL_dopacity", C.c_void_p), ("dL_dcolors", C.c_void_p),
        ("dL_dmeans3D", C.c_void_p), ("dL_dcov3D", C.c_void_p), ("dL_dsh", C.c_void_p), ("dL_dsh_rest", C.c_void_p),
        ("dL_dscales", C.c_void_p), ("dL_drotations", C.c_void_p), ("grad_accum_rezero", C.c_int32), ("num_units", C.c_int64),
        ("factor_campos_row", C.c_int32), ("sh_factor_mode", C.c_int32),
        # ABI 8: the mesh backward inside preprocess_bwd (frames rendered straight from a mesh)
        ("mesh", C.c_void_p), ("mesh_dL_dvertices", C.c_void_p), ("mesh_dL_dalpha", C.c_void_p), ("mesh_dL_dscale", C.c_void_p), ("mesh_dL_d_opacity", C.c_void_p),
    ]


class MeshArgs(C.Structure):
    _fields_ = [
        ("F", C.c_int32), ("V", C.c_int32), ("P", C.c_int64), ("splats_per_face", C.c_int32), ("alpha_mode", C.c_int32),
        ("vertices", C.c_void_p), ("faces", C.c_void_p), ("face_splat_offset", C.c_void_p), ("splat_face", C.c_void_p),
        ("_alpha", C.c_void_p), ("_scale", C.c_void_p), ("fused_activations", C.c_int32), ("_opacity", C.c_void_p),
        ("prezero", C.c_void_p), ("prezero_count", C.c_int64), ("vertex_grad_prezeroed", C.c_int32),
    ]


class LossArgs(C.Structure):
    _fields_ = [
        ("planes", C.c_int32), ("height", C.c_int32), ("width", C.c_int32), ("img", C.c_void_p), ("gt", C.c_void_p),
        ("w_l1", C.c_float), ("w_ssim", C.c_float), ("bias", C.c_float),
    ]


class ShGradExpandArgs(C.Structure):
    """GmsShGradExpandArgs (include/gmsplat.h)."""
    _fields_ = [
        ("P", C.c_int32), ("D", C.c_int32), ("M", C.c_int32), ("V", C.c_int32), ("means3D", C.c_void_p), ("campos", C.c_void_p),
        ("factors", C.c_void_p), ("factor_stride", C.c_int64), ("dL_dsh", C.c_void_p), ("dL_dsh_rest", C.c_void_p),
        ("accumulate", C.c_int32), ("debug", C.c_int32),
    ]


class AdamTensor(C.Structure):
    _fields_ = [
        ("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
        ("n", C.c_int64), ("lr", C.c_float), ("step", C.c_int32),
    ]


# every symbol include/gmsplat.h declares
EXPORTS = (
    "gms_rasterize_forward", "gms_rasterize_backward", "gms_mark_visible", "gms_mesh_to_gaussians_forward",
    "gms_mesh_to_gaussians_backward", "gms_rasterize_forward_counts", "gms_abi_version", "gms_last_error", "gms_geom_bytes", "gms_image_bytes",
    "gms_binning_bytes", "gms_profile_enable", "gms_profile_reset", "gms_profile_read", "gms_profile_kernel_name",
    "gms_knn_workspace_bytes", "gms_knn_mean_dist2", "gms_l1_ssim_partials", "gms_l1_ssim_forward",
    "gms_l1_ssim_backward", "gms_adam_step", "gms_wait_stats", "gms_last_deepest_tile", "gms_image_n_contrib_offset",
    "gms_sh_grad_expand", "gms_set_fault", "gms_get_fault", "gms_set_deterministic", "gms_get_deterministic",
    "gms_image_counts_offset", "gms_last_launched_units", "gms_last_used_micro",
    "gms_set_upstream_scale_mod_grad", "gms_get_upstream_scale_mod_grad", "gms_profile_event_overhead_us",
)
K_COUNT = 17

_lock = threading.Lock()
_lib = None


def load():
    """Load libgmsplat.so (once).  Raises RuntimeError with build instructions when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"libgmsplat.so not found at {LIB_PATH}: the HIP rasterizer is not built. "
                "Run `make -C gaussian-mesh-splatting_amd/csrc` (needs hipcc, targets gfx950). "
                "There is no CPU fallback in the product path.")
        lib = C.CDLL(LIB_PATH)
        lib.gms_rasterize_forward.restype = C.c_int64
        lib.gms_rasterize_forward.argtypes = [C.POINTER(RasterForwardArgs), C.c_void_p]
        lib.gms_rasterize_backward.restype = C.c_int32
        lib.gms_rasterize_backward.argtypes = [C.POINTER(RasterBackwardArgs), C.c_void_p]
        lib.gms_mark_visible.restype = C.c_int32
        lib.gms_mark_visible.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.gms_mesh_to_gaussians_forward.restype = C.c_int32
        lib.gms_mesh_to_gaussians_forward.argtypes = [C.POINTER(MeshArgs)] + [C.c_void_p] * 8
        lib.gms_mesh_to_gaussians_backward.restype = C.c_int32
        lib.gms_mesh_to_gaussians_backward.argtypes = [C.POINTER(MeshArgs)] + [C.c_void_p] * 9
        lib.gms_abi_version.restype = C.c_int32
        lib.gms_last_error.restype = C.c_char_p
        for n in ("gms_geom_bytes", "gms_image_bytes", "gms_binning_bytes", "gms_image_n_contrib_offset"):
            getattr(lib, n).restype = C.c_size_t
        lib.gms_geom_bytes.argtypes = [C.c_int32]
        lib.gms_image_bytes.argtypes = [C.c_int32, C.c_int32]
        lib.gms_image_n_contrib_offset.argtypes = [C.c_int32, C.c_int32]
        lib.gms_image_counts_offset.restype = C.c_size_t
        lib.gms_image_counts_offset.argtypes = [C.c_int32, C.c_int32]
        lib.gms_last_launched_units.restype = C.c_int64
        lib.gms_last_used_micro.restype = C.c_int32
        lib.gms_binning_bytes.argtypes = [C.c_int64, C.c_int32, C.c_int32]
        lib.gms_knn_workspace_bytes.restype = C.c_size_t
        lib.gms_knn_workspace_bytes.argtypes = [C.c_int32]
        lib.gms_knn_mean_dist2.restype = C.c_int32
        lib.gms_knn_mean_dist2.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.gms_l1_ssim_partials.restype = C.c_size_t
        lib.gms_l1_ssim_partials.argtypes = [C.c_int32, C.c_int32, C.c_int32]
        lib.gms_l1_ssim_forward.restype = C.c_int32
        lib.gms_l1_ssim_forward.argtypes = [C.POINTER(LossArgs)] + [C.c_void_p] * 4
        lib.gms_l1_ssim_backward.restype = C.c_int32
        lib.gms_l1_ssim_backward.argtypes = [C.POINTER(LossArgs)] + [C.c_void_p] * 4
        lib.gms_adam_step.restype = C.c_int32
        lib.gms_adam_step.argtypes = [C.POINTER(AdamTensor), C.c_int32, C.c_double, C.c_double, C.c_double, C.c_void_p]
        lib.gms_sh_grad_expand.restype = C.c_int32
        lib.gms_sh_grad_expand.argtypes = [C.POINTER(ShGradExpandArgs), C.c_void_p]
        lib.gms_last_deepest_tile.restype = C.c_int64
        lib.gms_wait_stats.restype = None
        lib.gms_wait_stats.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int32]
        lib.gms_profile_enable.argtypes = [C.c_int32]
        lib.gms_profile_enable.restype = None
        lib.gms_profile_reset.restype = None
        lib.gms_profile_read.argtypes = [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        lib.gms_profile_read.restype = C.c_int32
        lib.gms_profile_kernel_name.argtypes = [C.c_int32]
        lib.gms_profile_kernel_name.restype = C.c_char_p
        lib.gms_set_fault.argtypes = [C.c_int32]
        lib.gms_set_fault.restype = None
        lib.gms_get_fault.restype = C.c_int32
        lib.gms_profile_event_overhead_us.argtypes = [C.c_void_p, C.c_int32]
        lib.gms_profile_event_overhead_us.restype = C.c_double
        lib.gms_set_upstream_scale_mod_grad.argtypes = [C.c_int32]
        lib.gms_set_upstream_scale_mod_grad.restype = None
        lib.gms_get_upstream_scale_mod_grad.restype = C.c_int32
        lib.gms_set_deterministic.argtypes = [C.c_int32]
        lib.gms_set_deterministic.restype = None
        lib.gms_get_deterministic.restype = C.c_int32
        if lib.gms_abi_version() != GMS_ABI_VERSION:
            raise RuntimeError(f"libgmsplat.so ABI {lib.gms_abi_version()} != binding ABI {GMS_ABI_VERSION}; rebuild")
        _lib = lib
    return _lib


def kernel_times() -> dict:
    """{kernel name: (total_ms, launches)} since the last gms_profile_reset()."""
    lib = load()
    out = {}
    for k in range(K_COUNT):
        ms, n = C.c_double(0), C.c_int64(0)
        lib.gms_profile_read(k, C.byref(ms), C.byref(n))
        out[lib.gms_profile_kernel_name(k).decode()] = (ms.value, n.value)
    return out


def check(rc: int, what: str):
    if rc < 0:
        msg = load().gms_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed ({ERRORS.get(int(rc), rc)}): {msg}")
    return rc


class _NullGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NULL_GUARD = _NullGuard()


def on_device(device):
    """Context that makes `device` current for the HIP calls inside: free when it already is (the usual case),
    torch.cuda.device otherwise (its enter/exit costs ~10 us of host time per call)."""
    import torch
    return _NULL_GUARD if torch.cuda.current_device() == device.index else torch.cuda.device(device)


def stream_ptr(device):
    """Raw hipStream_t of torch's current stream on `device` (the ~1 us path; torch.cuda.current_stream() builds a
    Stream object and costs ~12 us)."""
    import torch
    return torch._C._cuda_getCurrentRawStream(device.index)


def ptr(t):
    """Device pointer of a tensor (None / empty -> NULL)."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and t.numel() and not t.is_cuda:
            raise RuntimeError(
                "diff_gaussian_rasterization (MI355X/HIP build): tensors must live on a GPU; "
                "there is no CPU path in the product (the CPU oracle lives under oracle/ for tests only)")
