"""GPU: the HIP kernels against golden vectors dumped from the UPSTREAM CUDA rasterizer (tests/golden/dump_upstream.py) -- BASELINE.json's
north_star tolerances, 1e-4 abs on rendered RGB and 1e-3 rel on gradients, in both reduction modes.  Skips ("parity unpinned") while no dump
is committed; the plumbing runs on every GPU pass against a dump of this repository's own drop-in written to a temporary directory."""
import os

import pytest

import _upstream as UP
import _util as U

pytestmark = pytest.mark.gpu


def _hip_side(inputs, kw, gc, gd):
    return U.hip_render(inputs, kw, grad_color=gc, grad_invdepth=gd)


@pytest.mark.skipif(not UP.dumps(), reason=UP.UNPINNED)
@pytest.mark.parametrize("path", UP.dumps() or ["-"])
def test_hip_kernels_equal_the_upstream_cuda_rasterizer(path):
    UP.compare(path, _hip_side, where="hip vs " + os.path.basename(path), both_modes=True)


def test_dump_script_runs_on_a_gpu_and_its_files_are_consumed(tmp_path):
    """The script as a CUDA-box user would run it (here with --self: the only rasterizer on this box is ours), then the comparison."""
    D = UP.dump_module()
    written = D.dump_all(str(tmp_path), self_module=True, device="cuda", only=["odd_size", "hotdog_slice_1k"])
    assert len(written) == 2
    for p in written:
        rep = UP.compare(p, _hip_side, where="gpu plumbing " + os.path.basename(p))
        assert rep["max_clean"] <= 1e-6          # the same kernels twice: the forward has no atomics
