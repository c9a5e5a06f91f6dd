"""CPU: the C oracle against golden vectors dumped from the UPSTREAM CUDA rasterizer (tests/golden/dump_upstream.py; SURVEY.md section
8(c) item 3).  This is the test that PINS the oracle for K1-K9 -- and it SKIPS, saying so, while no dump has been committed: the
submodule that holds the CUDA source is empty in the reference tree and this image has no CUDA toolchain.
The plumbing (dump -> npz -> load -> compare) is exercised on every run by dumping the oracle itself into a temporary directory."""
import os

import pytest
import torch

import _upstream as UP
import _util as U


def _oracle_side(inputs, kw, gc, gd):
    return U.oracle_render(inputs, kw, gc, gd)


@pytest.mark.skipif(not UP.dumps(), reason=UP.UNPINNED)
@pytest.mark.parametrize("path", UP.dumps() or ["-"])
def test_oracle_equals_the_upstream_cuda_rasterizer(path):
    UP.compare(path, _oracle_side, where="oracle vs " + os.path.basename(path))


def test_dump_load_compare_plumbing_on_the_oracle_itself(monkeypatch, tmp_path):
    """`dump_upstream.py --self` with the drop-in's kernel call served by the oracle (there is no GPU here): the file it writes loads back
    into the inputs / settings / outputs / gradients it was made from, and the comparison the real dumps will go through accepts it."""
    import diff_gaussian_rasterization as dgr
    from test_reference_render_cpu import _oracle_rasterize
    monkeypatch.setattr(dgr, "_rasterize_gaussians", _oracle_rasterize)
    D = UP.dump_module()
    written = D.dump_all(str(tmp_path), self_module=True, device="cpu", only=["deg0", "precomp"])
    assert [os.path.basename(p) for p in written] == ["upstream_deg0.npz", "upstream_precomp.npz"]
    inputs, kw, up, grads, prov = D.load(written[0])
    assert "self=True" in prov and kw["sh_degree"] == 0 and kw["image_width"] == 96 and kw["antialiasing"] is False
    assert set(inputs) == {"means3D", "opacities", "shs", "scales", "rotations"} and inputs["shs"].shape == (2000, 16, 3)
    assert up["color"].shape == (3, 96, 96) and up["radii"].dtype.kind == "i" and up["grad_color"].shape == (3, 96, 96)
    assert {"means3D", "means2D", "shs", "opacities", "scales", "rotations"} <= set(grads)
    for p in written:
        rep = UP.compare(p, _oracle_side, where="plumbing " + os.path.basename(p))
        assert rep["max_clean"] == 0.0                       # the oracle against a dump of itself
    # a dump whose gradients are off by 2e-3 does NOT pass: the comparison is the suite's criterion, not a formality
    import numpy as np
    z = dict(np.load(written[0]))
    z["dL_scales"] = z["dL_scales"] * (1.0 + 2e-3)
    bad = os.path.join(str(tmp_path), "upstream_bad.npz")
    np.savez_compressed(bad, **z)
    with pytest.raises(AssertionError):
        UP.compare(bad, _oracle_side, where="negative control")


def test_a_dump_of_this_repository_is_refused_without_self(tmp_path):
    D = UP.dump_module()
    with pytest.raises(SystemExit, match="pins nothing"):
        D.dump_all(str(tmp_path), self_module=False, device="cpu", only=["deg0"])
