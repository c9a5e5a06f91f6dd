"""Replacement for the reference's un-vendored `simple_knn` submodule (.gitmodules:1-3).

Every reference model imports `from simple_knn._C import distCUDA2` (scene/gaussian_model.py:20); `gs` / `gs_flat`
call it once at initialisation (scene/gaussian_model.py:134, games/flat_splatting/scene/flat_gaussian_model.py:47).
`_C.distCUDA2` is the HIP kernel set of csrc/knn.hip."""
