"""distCUDA2(points[N,3]) -> float[N]: mean squared distance to the 3 nearest neighbours (self excluded).

HIP implementation (csrc/knn.hip, uniform-grid exact search) behind `gms_knn_mean_dist2`; replaces the reference's
un-vendored `simple_knn._C.distCUDA2` (.gitmodules:1-3; call sites scene/gaussian_model.py:134,
games/flat_splatting/scene/flat_gaussian_model.py:47).  GPU tensors only, as upstream -- there is no CPU fallback."""
import ctypes as C

import torch

from diff_gaussian_rasterization import _lib


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    if points.dim() != 2 or points.shape[1] != 3:
        raise ValueError("distCUDA2 expects points of shape [N,3]")
    _lib.require_gpu(points)
    lib = _lib.load()
    pts = points.detach().to(torch.float32).contiguous()
    n = pts.shape[0]
    out = torch.empty(n, dtype=torch.float32, device=pts.device)
    if n == 0:
        return out
    with _lib.on_device(pts.device):
        nbytes = lib.gms_knn_workspace_bytes(n)
        work = torch.empty(nbytes, dtype=torch.uint8, device=pts.device)
        rc = lib.gms_knn_mean_dist2(n, _lib.ptr(pts), _lib.ptr(out), _lib.ptr(work), nbytes,
                                    C.c_void_p(_lib.stream_ptr(pts.device)))
    _lib.check(rc, "gms_knn_mean_dist2")
    return out
