"""Import the REAL reference python (from /root/reference) with stub modules for the absent
third-party dependencies.  Only usable in the authoring container: /root/reference does not
exist on the GPU box, so this module is used exclusively by tests/golden/make_golden.py (to
generate committed fixtures) and by CPU tests that skip when the tree is missing.

TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import contextlib
import os
import sys
import types

# GMS_REFERENCE_DIR (or the older GAMES_REFERENCE_ROOT): where a checkout of waczjoan/gaussian-mesh-splatting lives
REFERENCE_ROOT = os.environ.get("GMS_REFERENCE_DIR", os.environ.get("GAMES_REFERENCE_ROOT", "/root/reference"))


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "games"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    """Returns a namespace with the reference modules the hot path touches."""
    if not available():
        raise RuntimeError("reference tree not present")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # absent third-party deps: only their names are needed at import time
    try:            # the real package when installed, else the repo's stand-in (games_hip/_plyfile.py) under its name
        from games_hip._plyfile_compat import ensure_plyfile
        ensure_plyfile()
    except ImportError:
        _stub("plyfile", PlyData=object, PlyElement=object)
    for name in ("trimesh", "smplx", "smplx.lbs", "smplx.utils", "simple_knn", "simple_knn._C"):
        if name not in sys.modules:
            _stub(name)
    sys.modules["simple_knn._C"].distCUDA2 = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stub"))
    for n in ("lbs", "batch_rodrigues", "vertices2landmarks", "find_dynamic_lmk_idx_and_bcoords", "blend_shapes",
              "vertices2joints"):
        setattr(sys.modules["smplx.lbs"], n, None)
    for n in ("Struct", "to_tensor", "to_np", "rot_mat_to_euler"):
        setattr(sys.modules["smplx.utils"], n, None)
    if "diff_gaussian_rasterization" not in sys.modules:
        try:        # the drop-in itself when it is on sys.path (imports without a GPU); a name-only stub otherwise
            import diff_gaussian_rasterization  # noqa: F401
        except ImportError:
            _stub("diff_gaussian_rasterization", GaussianRasterizationSettings=None, GaussianRasterizer=None,
                  __games_stub__=True)
    import importlib

    ns = types.SimpleNamespace()
    ns.general_utils = importlib.import_module("utils.general_utils")
    ns.sh_utils = importlib.import_module("utils.sh_utils")
    ns.graphics_utils = importlib.import_module("utils.graphics_utils")
    ns.loss_utils = importlib.import_module("utils.loss_utils")
    ns.gaussian_model = importlib.import_module("scene.gaussian_model")
    ns.mesh_model = importlib.import_module("games.mesh_splatting.scene.gaussian_mesh_model")
    ns.multi_mesh_model = importlib.import_module("games.multi_mesh_splatting.scene.gaussian_multi_mesh_model")
    ns.flame_model = importlib.import_module("games.flame_splatting.scene.gaussian_flame_model")
    return ns


def drop_reference_stubs():
    """Remove the `diff_gaussian_rasterization` stub so the real drop-in can be imported afterwards."""
    m = sys.modules.get("diff_gaussian_rasterization")
    if m is not None and getattr(m, "__games_stub__", False):
        del sys.modules["diff_gaussian_rasterization"]


@contextlib.contextmanager
def cuda_literals_on_cpu():
    """The reference hard-codes device="cuda" / .cuda() (utils/general_utils.py:145,163,182, every create_from_pcd).
    Run its code unmodified on CPU by making tensor factories ignore that literal and `.cuda()` a no-op for the
    duration."""
    import torch

    names = ("zeros", "ones", "tensor", "empty", "zeros_like", "ones_like", "rand", "full")
    orig = {n: getattr(torch, n) for n in names}

    def wrap(fn):
        def inner(*a, **k):
            if k.get("device") == "cuda":
                k = dict(k)
                k["device"] = "cpu"
            return fn(*a, **k)
        return inner

    orig_cuda = torch.Tensor.cuda
    orig_mod_cuda = torch.nn.Module.cuda
    orig_to = torch.Tensor.to

    def to(self, *a, **k):          # `.to("cuda")` (train.py:204)
        if a and isinstance(a[0], str) and a[0].startswith("cuda"):
            a = ("cpu",) + tuple(a[1:])
        if isinstance(k.get("device"), str) and k["device"].startswith("cuda"):
            k = dict(k, device="cpu")
        return orig_to(self, *a, **k)

    for n in names:
        setattr(torch, n, wrap(orig[n]))
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.Tensor.to = to
    try:
        yield
    finally:
        for n in names:
            setattr(torch, n, orig[n])
        torch.Tensor.cuda = orig_cuda
        torch.nn.Module.cuda = orig_mod_cuda
        torch.Tensor.to = orig_to
