"""GPU parity of simple_knn._C.distCUDA2 (csrc/knn.hip) against the exact-3NN oracle (oracle/knn_oracle.py).
Tolerance: 1e-5 relative (+1e-12 abs): the result is a mean of three float32 squared distances; the only freedom is
fused-multiply-add contraction inside dx*dx+dy*dy+dz*dz."""
import numpy as np
import pytest
import torch

from oracle import knn_oracle

pytestmark = pytest.mark.gpu


def _check(points, ref):
    from simple_knn._C import distCUDA2
    got = distCUDA2(torch.from_numpy(points).cuda()).cpu().numpy()
    assert got.shape == (points.shape[0],) and got.dtype == np.float32
    err = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-12)
    bad = (np.abs(got - ref) > 1e-5 * np.abs(ref) + 1e-12)
    assert not bad.any(), (int(bad.sum()), float(err.max()), got[bad][:4], ref[bad][:4])


def _clouds():
    rng = np.random.default_rng(7)
    uniform = rng.uniform(-1, 1, size=(5000, 3)).astype(np.float32)
    clustered = np.concatenate([rng.normal(c, s, size=(1500, 3)) for c, s in
                                [((0, 0, 0), 0.01), ((5, 5, 5), 1.0), ((-40, 2, 9), 0.2)]]).astype(np.float32)
    planar = uniform.copy(); planar[:, 2] = 0.25
    collinear = np.zeros((2000, 3), np.float32); collinear[:, 0] = rng.uniform(0, 10, 2000)
    dup = uniform[:1000].copy(); dup[100:400] = dup[0]; dup[500:502] = dup[499]
    outlier = uniform[:2000].copy(); outlier[17] = (1e4, -3e3, 50.0)
    lattice = np.stack(np.meshgrid(*[np.arange(12, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(-1, 3)
    return dict(uniform=uniform, clustered=clustered, planar=planar, collinear=collinear, duplicates=dup,
                outlier=outlier, lattice=lattice)


@pytest.mark.parametrize("name", list(_clouds().keys()))
def test_matches_bruteforce(name):
    p = _clouds()[name]
    _check(p, knn_oracle.dist2_bruteforce(p))


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 5, 63, 64, 65])
def test_tiny_clouds(n):
    p = np.random.default_rng(n).normal(size=(n, 3)).astype(np.float32)
    _check(p, knn_oracle.dist2_bruteforce(p))


def test_single_repeated_point():
    p = np.full((300, 3), 1.5, np.float32)
    _check(p, np.zeros(300, np.float32))


def test_full_size_sfm_like_cloud_matches_kdtree():
    """300k points (the headline scene's Gaussian count) on a noisy sphere shell + background: exact k-d tree check."""
    rng = np.random.default_rng(3)
    d = rng.normal(size=(300_000, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    p = (d * (1.0 + 0.01 * rng.normal(size=(300_000, 1)))).astype(np.float32)
    p[:20_000] = rng.uniform(-8, 8, size=(20_000, 3)).astype(np.float32)
    _check(p, knn_oracle.dist2_kdtree(p).astype(np.float32))


def test_rejects_cpu_tensors_and_bad_shapes():
    from simple_knn._C import distCUDA2
    with pytest.raises(RuntimeError):
        distCUDA2(torch.zeros(8, 3))
    with pytest.raises(ValueError):
        distCUDA2(torch.zeros(8, 2, device="cuda"))


def test_reference_init_expression():
    """scene/gaussian_model.py:134-135: scales = log(sqrt(clamp_min(distCUDA2(points), 1e-7))) stays finite."""
    from simple_knn._C import distCUDA2
    p = torch.from_numpy(_clouds()["duplicates"]).cuda()
    s = torch.log(torch.sqrt(torch.clamp_min(distCUDA2(p), 0.0000001)))
    assert torch.isfinite(s).all()
