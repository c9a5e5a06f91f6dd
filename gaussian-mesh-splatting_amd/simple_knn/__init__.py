"""Import-time stand-in for the reference's un-vendored `simple_knn` submodule (.gitmodules:1-3).

Every reference model imports `from simple_knn._C import distCUDA2` (scene/gaussian_model.py:20) even
though the mesh models never call it.  This package keeps those imports working.  `distCUDA2` here is a
plain PyTorch (device-agnostic, exact, chunked) implementation used only at initialisation of `gs` /
`gs_flat` (scene/gaussian_model.py:134, games/flat_splatting/scene/flat_gaussian_model.py:47); a native
HIP Morton-sort kNN is SURVEY.md section 8(f) item 1 and is NOT part of the hot path delivered here."""
