"""CPU (hipcc cross-compiles gfx950 here): what the kernels ask of a CU, from the compiler's own resource remarks
(tools/kernel_resources.py).  No kernel may spill or use scratch, and the occupancies DESIGN.md sections 4 and 7 argue from are
the ones the compiler reports for the shipped instantiations."""
import os
import shutil
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not installed")


@pytest.fixture(scope="module")
def table():
    import kernel_resources as kr
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for f in ("blend_micro.hip", "raster_backward.hip", "mesh_to_gaussians.hip"):
            ks = kr.remarks(f, tmp)
            for k, n in zip(ks, kr.demangle([k["name"] for k in ks])):
                out[n] = k
    return out


def test_no_kernel_spills_or_uses_scratch(table):
    assert len(table) > 25
    for n, k in table.items():
        assert k.get("scratch", 0) == 0 and k.get("vspill", 0) == 0 and k.get("sspill", 0) == 0, (n, k)
        assert k.get("agpr", 0) == 0, (n, k)                      # no MFMA anywhere on this path: no accumulation registers either


def test_occupancies_design_md_argues_from(table):
    bwd = table["micro_bwd_kernel<false, 2, 0, false, true, 256>"]           # the shipped backward: 64-bit fixed-point table
    assert bwd["lds"] <= 31984 and bwd["occ"] == 5                      # five blocks per CU
    det = table["micro_bwd_kernel<false, 2, 0, true, true, 256>"]
    assert det["occ"] == 5 and det["lds"] == bwd["lds"]                 # the deterministic mode rides on the same table
    flt = table["micro_bwd_kernel<false, 2, 0, false, false, 256>"]          # GMS_BWD_FIXED=0: the float table, six blocks
    assert flt["occ"] == 6 and flt["lds"] < bwd["lds"]
    for name in ("micro_head_kernel<4>", "micro_fwd_kernel<4>"):        # forward compositing: at the wave limit
        assert table[name]["occ"] == 8 and table[name]["vgpr"] <= 64, name
    pre = table["preprocess_bwd_kernel<false>"]
    assert pre["occ"] == 3 and pre["lds"] == 53248                      # 52 KB of SH rows: three blocks per CU
    fused = table["preprocess_bwd_kernel<true>"]                         # ... with the mesh backward in its tail (ABI 8): the same three
    assert fused["occ"] == 3 and fused["lds"] == pre["lds"] and fused.get("scratch", 0) == 0
    assert table["mesh_bwd_fused_kernel"]["occ"] == 5                   # (4 with the SLP vectoriser's packed operands: round 6, Makefile NOSLP)
