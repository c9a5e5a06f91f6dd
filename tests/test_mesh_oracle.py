"""K0 oracle (oracle/mesh_oracle.py) against fixtures produced by EXECUTING the reference's own
GaussianMeshModel / GaussianMultiMeshModel (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import mesh_oracle


def _load(golden_dir, name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, name)).items()}


def _loss(xyz, scaling, rotation, g):
    return ((xyz * g["g_xyz"]).sum() + (torch.exp(scaling) * g["g_scaling_act"]).sum()
            + (torch.nn.functional.normalize(rotation) * g["g_rotation_act"]).sum())


@pytest.mark.parametrize("name", ["k0_mesh.npz", "k0_mesh_s5.npz"])
def test_single_mesh_matches_reference(golden_dir, name):
    g = _load(golden_dir, name)
    v = g["vertices"].clone().requires_grad_(True)
    a = g["_alpha"].clone().requires_grad_(True)
    s = g["_scale"].clone().requires_grad_(True)
    alpha, tri, xyz, scaling, rot = mesh_oracle.mesh_to_gaussians(v, g["faces"], a, s, "relu")
    # forward: same op sequence on the same machine -> bit-exact
    for got, key in ((alpha, "alpha"), (tri, "triangles"), (xyz, "xyz"), (scaling, "scaling"), (rot, "rotation")):
        assert torch.equal(got.detach(), g[key]), key
    _loss(xyz, scaling, rot, g).backward()
    torch.testing.assert_close(v.grad, g["d_vertices"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(a.grad, g["d_alpha"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(s.grad, g["d_scale"], rtol=1e-5, atol=1e-6)


def test_multi_mesh_matches_reference(golden_dir):
    g = _load(golden_dir, "k0_multi_mesh.npz")
    vs = [g[f"vertices{i}"].clone().requires_grad_(True) for i in range(2)]
    al = [g[f"_alpha{i}"].clone().requires_grad_(True) for i in range(2)]
    sc = [g[f"_scale{i}"].clone().requires_grad_(True) for i in range(2)]
    xyz, scaling, rot = mesh_oracle.multi_mesh_to_gaussians(vs, [g[f"faces{i}"] for i in range(2)], al, sc)
    assert torch.equal(xyz.detach(), g["xyz"])
    assert torch.equal(scaling.detach(), g["scaling"])
    assert torch.equal(rot.detach(), g["rotation"])
    _loss(xyz, scaling, rot, g).backward()
    for i in range(2):
        torch.testing.assert_close(vs[i].grad, g[f"d_vertices{i}"], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(al[i].grad, g[f"d_alpha{i}"], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(sc[i].grad, g[f"d_scale{i}"], rtol=1e-5, atol=1e-6)


def test_flame_variant_matches_reference(golden_dir):
    """softmax alpha + vertices through the FLAME transform (gaussian_flame_model.py:123-207), fixture = the reference's
    GaussianFlameModel executed with a stubbed FLAME layer."""
    g = _load(golden_dir, "k0_flame.npz")
    v0 = g["flame_vertices"].clone().requires_grad_(True)
    enl = g["enlargement"].clone().requires_grad_(True)
    vv = torch.squeeze(v0[None])
    verts = torch.stack([vv[:, 0], -vv[:, 2], vv[:, 1]], dim=1) * enl       # dataset_readers.py:41-46
    a = g["_alpha"].clone().requires_grad_(True)
    s = g["_scales"].clone().requires_grad_(True)
    alpha, _, xyz, scaling, rot = mesh_oracle.mesh_to_gaussians(verts, g["faces"], a, s, "softmax")
    assert torch.equal(verts.detach(), g["vertices"])
    for got, key in ((alpha, "alpha"), (xyz, "xyz"), (scaling, "scaling"), (rot, "rotation")):
        assert torch.equal(got.detach(), g[key]), key
    _loss(xyz, scaling, rot, g).backward()
    torch.testing.assert_close(v0.grad, g["d_flame_vertices"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(enl.grad, g["d_enlargement"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(a.grad, g["d_alpha"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(s.grad, g["d_scales"], rtol=1e-5, atol=1e-6)


def test_quaternion_of_rotation_matches_reference(golden_dir):
    g = _load(golden_dir, "stages.npz")
    assert torch.equal(mesh_oracle.rot_to_quat_batch(g["R"]), g["quat_of_R"])


def test_frame_is_right_handed_and_scales_positive():
    from games_hip import synthetic as syn
    v, f = syn.uv_sphere(10, 12)
    a = torch.rand(f.shape[0], 2, 3)
    s = torch.ones(f.shape[0] * 2, 1)
    _, tri, xyz, scaling, rot = mesh_oracle.mesh_to_gaussians(v, f, a, s)
    v0, v1, v2, _ = mesh_oracle.face_frames(tri)
    det = torch.linalg.det(torch.stack((v0, v1, v2), dim=1))
    assert torch.allclose(det, torch.ones_like(det), atol=1e-4)
    assert torch.allclose(rot.norm(dim=1), torch.ones(rot.shape[0]), atol=1e-5)
    # s0 ends up 2e-8 (relu(1*1e-8)+1e-8)
    assert torch.allclose(torch.exp(scaling[:, 0]), torch.full((scaling.shape[0],), 2e-8), rtol=1e-5)
    # centres lie inside their triangles' bounding boxes
    P = xyz.reshape(f.shape[0], 2, 3)
    assert (P <= tri.max(dim=1).values[:, None] + 1e-6).all() and (P >= tri.min(dim=1).values[:, None] - 1e-6).all()


def test_softmax_alpha_mode():
    from games_hip import synthetic as syn
    v, f = syn.uv_sphere(6, 6)
    a = torch.randn(f.shape[0], 4, 3)
    alpha, tri, xyz, _, _ = mesh_oracle.mesh_to_gaussians(v, f, a, torch.ones(f.shape[0] * 4, 1), "softmax")
    assert torch.allclose(alpha.sum(-1), torch.ones(f.shape[0], 4), atol=1e-6)
    assert torch.allclose(xyz, torch.matmul(torch.softmax(a, 2), v[f]).reshape(-1, 3))
