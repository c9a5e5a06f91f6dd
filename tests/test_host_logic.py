"""Host-side logic that needs no GPU: the capacity-hint grid of the binning buffer and the source hash that ties PMC
counters to a build."""
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_capacity_hints_sit_on_a_coarse_monotone_grid():
    """diff_gaussian_rasterization._quantize_capacity (same rule as torch_binding.cpp::quantize_capacity): never below the
    request, at most 12.5 % above it, monotone, idempotent -- and a request that creeps up by 0.5 % per frame (an animated
    mesh) changes the buffer size once per ~12-25 frames instead of every frame."""
    from diff_gaussian_rasterization import _quantize_capacity as q
    xs = np.unique(np.concatenate([np.arange(1, 5000), np.random.default_rng(0).integers(5000, 1 << 40, 20000)]))
    qs = np.array([q(int(x)) for x in xs], dtype=np.float64)
    assert (qs >= xs).all() and (qs <= np.maximum(xs * 1.125, xs + 1)).all()
    assert (np.diff(qs) >= 0).all()
    assert all(q(int(v)) == int(v) for v in qs[::97])
    n, sizes = 11_760_000, set()
    for _ in range(125):
        sizes.add(q(int(n * 1.25) + 4096))
        n = int(n * 1.005)
    assert len(sizes) <= 12, sorted(sizes)


def test_kernel_source_hash_ignores_comments_but_not_code(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import srchash
    finally:
        sys.path.pop(0)
    src = os.path.join(ROOT, "gaussian-mesh-splatting_amd", "csrc")
    dst = tmp_path / "csrc"
    dst.mkdir()
    for name in os.listdir(src):
        if name.endswith((".hip", ".h")):
            shutil.copy(os.path.join(src, name), dst / name)
    h0 = srchash.kernel_source_hash(str(dst))
    assert h0 == srchash.kernel_source_hash(src)
    f = dst / "blend_micro.hip"
    text = f.read_text()
    f.write_text("// a new remark\n" + text.replace("\n", "\n   \n", 3) + "\n/* block\n comment */\n")
    assert srchash.kernel_source_hash(str(dst)) == h0
    f.write_text(text.replace("constexpr int LMAX = 256;", "constexpr int LMAX = 128;"))
    assert "LMAX = 128" in f.read_text() and srchash.kernel_source_hash(str(dst)) != h0


def test_committed_pmc_counters_belong_to_the_committed_kernels():
    """profiles/pmc_traffic.json is only reported by bench.py when its `_source_hash` equals the hash of the sources being run
    (regenerate with tools/r04_final.sh (collect_profiles.sh + make_pmc_traffic.py) after a kernel change); a stale pair shows up here as a SKIP."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import srchash
    finally:
        sys.path.pop(0)
    import pytest
    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    have, want = pmc["c2_hotdog_like/trained"]["_source_hash"], srchash.kernel_source_hash()
    assert isinstance(have, str) and len(have) == 16
    if have != want:        # not an error (bench.py withholds stale counters by itself), but say so where a developer sees it
        pytest.skip(f"profiles/pmc_traffic.json is stale (collected on {have}, sources are {want}): bench.py reports traffic = null")
