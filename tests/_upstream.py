"""Shared by tests/test_oracle_upstream_golden.py (CPU) and tests/test_gpu_upstream_golden.py (GPU): compare one side -- the C oracle or
the HIP kernels -- with a dump of the UPSTREAM CUDA rasterizer (tests/golden/dump_upstream.py), under BASELINE.json's tolerances (rendered
RGB <= 1e-4 abs, gradients <= 1e-3 rel) and the suite's discontinuity rule (tests/test_gpu_raster.py: pixels / Gaussians whose skip, stop,
radius or rectangle decision lies within float rounding are flagged by the oracle and compared with the bound a flipped decision implies)."""
import glob
import importlib.util
import os

import numpy as np

import _util as U

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
UNPINNED = ("parity unpinned: no tests/golden/upstream_*.npz -- run tests/golden/dump_upstream.py on a machine that has the upstream CUDA "
            "diff_gaussian_rasterization built and commit the files (SURVEY.md section 8(c) item 3)")


def dumps():
    return sorted(glob.glob(os.path.join(GOLDEN, "upstream_*.npz")))


def dump_module():
    spec = importlib.util.spec_from_file_location("dump_upstream", os.path.join(GOLDEN, "dump_upstream.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def compare(path, render, where, both_modes=False):
    """`render(inputs, kw, grad_color, grad_invdepth)` -> dict(color, radii, invdepth, grads) of the side under test.  The dump plays the
    role the float32 oracle plays in the suite; the oracle still supplies the ambiguity flags, the float64 run and the realisations."""
    D = dump_module()
    inputs, kw, up, up_grads, prov = D.load(path)
    W, H = kw["image_width"], kw["image_height"]
    gc, gd = up["grad_color"], up["grad_invdepth"]
    o = U.oracle_render(inputs, kw, gc, gd)
    ref = dict(color=up["color"], radii=up["radii"], invdepth=up.get("invdepth", o["invdepth"]), details=o["details"])
    side = render(inputs, kw, gc, gd)
    rep = U.forward_report(side, ref, W, H)
    assert rep["radii_unexplained"] == 0, (where, prov, rep)
    assert rep["amb_frac"] < 0.01 and rep["max_clean"] <= 1e-4 and rep["max_amb"] <= 0.02, (where, prov, rep)
    if "invdepth" in up:
        assert rep["max_invdepth_clean"] <= 1e-4, (where, prov, rep)
    go = {k: up_grads.get(k) for k in ("means3D", "means2D", "shs", "colors_precomp", "opacities", "scales", "rotations", "cov3D_precomp")}
    args = dict(go64_fn=lambda: U.oracle_render(inputs, kw, gc, gd, precision="f64")["grads"], where=where, excuse=U.excused_rows(o["details"]),
                go32acc_fn=lambda: U.f32_realisations(inputs, kw, gc, gd), alt=U.alt_oracles(inputs, kw, gc, gd, o["details"]))
    if both_modes:
        U.assert_grads_both_modes(lambda: render(inputs, kw, gc, gd), go, **args)
    else:
        U.assert_grads(side["grads"], go, **args)
    return rep
