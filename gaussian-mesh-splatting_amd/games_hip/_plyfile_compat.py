"""`PlyData` / `PlyElement`: the third-party `plyfile` package when it is installed, otherwise the numpy stand-in
(`games_hip/_plyfile.py`: the calls the reference makes, PLY 1.0 ascii / binary).  `ensure_plyfile()` registers the
stand-in under the name `plyfile` ONLY when the real package is missing, so `from plyfile import PlyData, PlyElement`
in the reference (scene/gaussian_model.py:19, scene/dataset_readers.py:22) resolves either way and a real install is
never shadowed."""
import sys

try:
    from plyfile import PlyData, PlyElement  # type: ignore  # noqa: F401
    REAL = True
except ImportError:
    from ._plyfile import PlyData, PlyElement  # noqa: F401
    REAL = False


def ensure_plyfile():
    if "plyfile" not in sys.modules:
        try:
            import plyfile  # type: ignore  # noqa: F401
        except ImportError:
            from . import _plyfile
            sys.modules["plyfile"] = _plyfile
    return sys.modules["plyfile"]
