"""GPU parity of the fused mesh->Gaussian op (K0) against fixtures produced by executing the
reference's own GaussianMeshModel / GaussianMultiMeshModel, and against the torch restatement."""
import os

import numpy as np
import pytest
import torch

from games_hip import synthetic as syn
from oracle import mesh_oracle

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, name)).items()}


def _loss(xyz, scaling, rotation, g, dev):
    return ((xyz * g["g_xyz"].to(dev)).sum() + (torch.exp(scaling) * g["g_scaling_act"].to(dev)).sum()
            + (torch.nn.functional.normalize(rotation) * g["g_rotation_act"].to(dev)).sum())


def _close(a, b, rtol=2e-4, atol_rel=2e-5):
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    atol = atol_rel * float(b.abs().max()) + 1e-12
    bad = (a - b).abs() > rtol * b.abs() + atol
    assert not bad.any(), (float((a - b).abs().max()), float(b.abs().max()), int(bad.sum()))


@pytest.mark.parametrize("name", ["k0_mesh.npz", "k0_mesh_s5.npz"])
def test_single_mesh_matches_reference_fixture(golden_dir, name):
    from games_hip.mesh_op import mesh_to_gaussians
    g = _load(golden_dir, name)
    v = g["vertices"].cuda().requires_grad_(True)
    a = g["_alpha"].cuda().requires_grad_(True)
    s = g["_scale"].cuda().requires_grad_(True)
    alpha, xyz, scaling, rot = mesh_to_gaussians(v, g["faces"].cuda(), a, s, "relu")
    _close(alpha, g["alpha"], rtol=1e-6)
    _close(xyz, g["xyz"], rtol=1e-5)
    # degenerate faces (zero area) give garbage-but-finite frames in the reference: exclude their splats
    tri = g["triangles"]
    area = torch.linalg.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]).norm(dim=1)
    S = g["_alpha"].shape[1]
    ok = (area > 1e-9).repeat_interleave(S)
    _close(scaling[ok.cuda()], g["scaling"][ok], rtol=1e-5)
    _close(rot[ok.cuda()], g["rotation"][ok], rtol=1e-4)
    assert torch.isfinite(rot).all() and torch.isfinite(scaling).all()
    if ok.all():
        _loss(xyz, scaling, rot, g, "cuda").backward()
        _close(v.grad, g["d_vertices"])
        _close(a.grad, g["d_alpha"])
        _close(s.grad, g["d_scale"])


def test_multi_mesh_csr_matches_reference_fixture(golden_dir):
    from games_hip.mesh_op import mesh_to_gaussians
    g = _load(golden_dir, "k0_multi_mesh.npz")
    verts, faces, alphas, scales, offs, sf = [], [], [], [], [0], []
    voff = foff = 0
    for i in range(2):
        v, f, a, s = g[f"vertices{i}"], g[f"faces{i}"], g[f"_alpha{i}"], g[f"_scale{i}"]
        verts.append(v); faces.append(f + voff); alphas.append(a.reshape(-1, 3)); scales.append(s)
        S = a.shape[1]
        for k in range(f.shape[0]):
            offs.append(offs[-1] + S)
            sf += [foff + k] * S
        voff += v.shape[0]; foff += f.shape[0]
    V = torch.cat(verts).cuda().requires_grad_(True)
    A = torch.cat(alphas).cuda().requires_grad_(True)
    Sc = torch.cat(scales).cuda().requires_grad_(True)
    _, xyz, scaling, rot = mesh_to_gaussians(V, torch.cat(faces).cuda(), A, Sc, "relu",
                                             face_splat_offset=torch.tensor(offs, dtype=torch.int32).cuda(),
                                             splat_face=torch.tensor(sf, dtype=torch.int32).cuda())
    _close(xyz, g["xyz"], rtol=1e-5); _close(scaling, g["scaling"], rtol=1e-5); _close(rot, g["rotation"], rtol=1e-4)
    _loss(xyz, scaling, rot, g, "cuda").backward()
    _close(V.grad, torch.cat([g["d_vertices0"], g["d_vertices1"]]))
    _close(A.grad, torch.cat([g["d_alpha0"].reshape(-1, 3), g["d_alpha1"].reshape(-1, 3)]))
    _close(Sc.grad, torch.cat([g["d_scale0"], g["d_scale1"]]))


@pytest.mark.parametrize("mode,S", [("relu", 3), ("softmax", 4), ("softmax", 50), ("relu", 100)])
def test_against_torch_restatement(mode, S):
    """Thread-per-face (S < 16) and wave-per-face (S >= 16) backward paths, both alpha modes."""
    from games_hip.mesh_op import mesh_to_gaussians
    v, f = syn.uv_sphere(12, 14)
    gen = torch.Generator().manual_seed(S)
    F = f.shape[0]
    a = torch.randn(F, S, 3, generator=gen) if mode == "softmax" else torch.rand(F, S, 3, generator=gen) - 0.1
    s = torch.exp(0.3 * torch.randn(F * S, 1, generator=gen))
    gx, gs, gr = torch.randn(F * S, 3, generator=gen), torch.randn(F * S, 3, generator=gen), torch.randn(F * S, 4, generator=gen)
    g = dict(g_xyz=gx, g_scaling_act=gs, g_rotation_act=gr)
    vc, ac, sc = v.clone().requires_grad_(True), a.clone().requires_grad_(True), s.clone().requires_grad_(True)
    _, _, xyz, scaling, rot = mesh_oracle.mesh_to_gaussians(vc, f, ac, sc, mode)
    _loss(xyz, scaling, rot, g, "cpu").backward()
    vg, ag, sg = v.cuda().requires_grad_(True), a.cuda().requires_grad_(True), s.cuda().requires_grad_(True)
    alpha_h, xyz_h, scaling_h, rot_h = mesh_to_gaussians(vg, f.cuda(), ag, sg, mode)
    _loss(xyz_h, scaling_h, rot_h, g, "cuda").backward()
    _close(xyz_h, xyz, rtol=1e-5); _close(scaling_h, scaling, rtol=1e-5); _close(rot_h, rot, rtol=1e-4)
    _close(vg.grad, vc.grad); _close(ag.grad, ac.grad); _close(sg.grad, sc.grad)


def test_triangles_mode_and_model_surface():
    """The animated renderer replaces pc.triangles and calls prepare_scaling_rot()
    (renderer/gaussian_animated_renderer/__init__.py:61-73): scales/rotations must follow the new triangles."""
    from games_hip.model import HipGaussianMeshModel
    scene = syn.mesh_scene("tiny")
    m = HipGaussianMeshModel.from_scene(scene, "cuda")
    assert m.get_xyz.shape == (scene.num_gaussians, 3) and m.alpha.shape == scene._alpha.shape
    new_v = scene.vertices * torch.tensor([1.0, 1.3, 0.8])
    tri = new_v[scene.faces].cuda()
    with torch.no_grad():
        m.triangles = tri
        m.prepare_scaling_rot()
        xyz = torch.matmul(m.alpha, tri).reshape(-1, 3)
    _, _, xyz_o, scaling_o, rot_o = mesh_oracle.mesh_to_gaussians(new_v, scene.faces, scene._alpha, scene._scale)
    _close(xyz, xyz_o, rtol=1e-5); _close(m._scaling, scaling_o, rtol=1e-5); _close(m._rotation, rot_o, rtol=1e-4)


@pytest.mark.parametrize("S", [3, 40])
def test_fused_activations_match_unfused_chain(S):
    """fused_activations=True returns exp(_scaling) / normalize(_rotation) and differentiates through them
    exactly like the reference's property getters (scene/gaussian_model.py:95-101) applied afterwards."""
    from games_hip.mesh_op import mesh_to_gaussians
    v, f = syn.uv_sphere(10, 12)
    gen = torch.Generator().manual_seed(S)
    F = f.shape[0]
    a = torch.rand(F, S, 3, generator=gen) - 0.05
    s = torch.exp(0.3 * torch.randn(F * S, 1, generator=gen))
    g = dict(g_xyz=torch.randn(F * S, 3, generator=gen), g_scaling_act=torch.randn(F * S, 3, generator=gen),
             g_rotation_act=torch.randn(F * S, 4, generator=gen))
    vc, ac, sc = v.clone().requires_grad_(True), a.clone().requires_grad_(True), s.clone().requires_grad_(True)
    _, _, xyz, scaling, rot = mesh_oracle.mesh_to_gaussians(vc, f, ac, sc, "relu")
    _loss(xyz, scaling, rot, g, "cpu").backward()
    vg, ag, sg = v.cuda().requires_grad_(True), a.cuda().requires_grad_(True), s.cuda().requires_grad_(True)
    _, xyz_h, scaling_h, rot_h, sact, runit = mesh_to_gaussians(vg, f.cuda(), ag, sg, "relu", fused_activations=True)
    _close(sact, torch.exp(scaling), rtol=1e-5)
    _close(runit, torch.nn.functional.normalize(rot), rtol=1e-4)
    _close(scaling_h, scaling, rtol=1e-5)
    ((xyz_h * g["g_xyz"].cuda()).sum() + (sact * g["g_scaling_act"].cuda()).sum() + (runit * g["g_rotation_act"].cuda()).sum()).backward()
    _close(vg.grad, vc.grad); _close(ag.grad, ac.grad); _close(sg.grad, sc.grad)


def test_multi_mesh_mixin_matches_reference_fixture(golden_dir):
    """Model-level drop-in for GaussianMultiMeshModel (lists of per-mesh tensors, different splats per face)."""
    from games_hip.model import HipMultiMeshMixin
    g = _load(golden_dir, "k0_multi_mesh.npz")

    class M(HipMultiMeshMixin):
        pass
    m = M()
    m.vertices = [g[f"vertices{i}"].cuda().requires_grad_(True) for i in range(2)]
    m.faces = [g[f"faces{i}"].cuda() for i in range(2)]
    m._alpha = [g[f"_alpha{i}"].cuda().requires_grad_(True) for i in range(2)]
    m._scale = [g[f"_scale{i}"].cuda().requires_grad_(True) for i in range(2)]
    m.update_alpha()
    m.prepare_scaling_rot()
    _close(m._xyz, g["xyz"], rtol=1e-5); _close(m._scaling, g["scaling"], rtol=1e-5); _close(m._rotation, g["rotation"], rtol=1e-4)
    assert [tuple(a.shape) for a in m.alpha] == [tuple(a.shape) for a in m._alpha]
    ((m._xyz * g["g_xyz"].cuda()).sum() + (m.get_scaling * g["g_scaling_act"].cuda()).sum()
     + (m.get_rotation * g["g_rotation_act"].cuda()).sum()).backward()
    for i in range(2):
        _close(m.vertices[i].grad, g[f"d_vertices{i}"]); _close(m._alpha[i].grad, g[f"d_alpha{i}"]); _close(m._scale[i].grad, g[f"d_scale{i}"])


def test_animated_render_follows_the_deformed_mesh():
    """BASELINE config 5 shape (per-frame vertex animation + on-device R/S re-derivation), small: the HIP
    animated render equals the oracle rendering of the oracle-derived Gaussians of the deformed mesh."""
    import _util as U
    from games_hip.model import HipGaussianMeshModel
    from games_hip.render import PipelineParams, render_animated
    from oracle import gs_oracle
    scene = syn.mesh_scene("small")
    model = HipGaussianMeshModel.from_scene(scene, "cuda")
    cam = syn.orbit_camera(2, width=128, height=128)
    t = 0.7
    new_v = scene.vertices * torch.tensor([1.0 + 0.2 * t, 1.0, 1.0 - 0.1 * t]) + torch.tensor([0.0, 0.05 * t, 0.0])
    with torch.no_grad():
        img = render_animated(None, new_v[scene.faces].cuda(), cam.to("cuda"), model, PipelineParams(), torch.ones(3, device="cuda"))["render"]
    _, _, xyz, scaling, rot = mesh_oracle.mesh_to_gaussians(new_v, scene.faces, scene._alpha, scene._scale)
    xyz_a, s_a, r_a, o_a, shs = mesh_oracle.activated(xyz, scaling, rot, scene._opacity, scene._features_dc, scene._features_rest)
    kw = {k: v for k, v in U.settings_kwargs(cam, torch.ones(3)).items() if k not in ("prefiltered", "debug")}
    o = gs_oracle.rasterize(means3D=xyz_a, opacities=o_a, shs=shs, scales=s_a, rotations=r_a, **kw)
    amb = o.state.details()["pix_ambig"].astype(bool)
    diff = np.abs(img.cpu().numpy() - o.color).max(0)
    assert (diff[~amb] <= 2e-4).mean() > 0.999 and amb.mean() < 0.02      # K0 rounding can move a radius by one


def test_fused_opacity_getter_matches_torch_sigmoid_and_its_gradient():
    """get_opacity = sigmoid(_opacity) (scene/gaussian_model.py:113-115) rides in the K0 kernels."""
    from games_hip.mesh_op import mesh_to_gaussians
    scene = syn.mesh_scene("tiny")
    v, f = scene.vertices.cuda().requires_grad_(True), scene.faces.cuda()
    a, s = scene._alpha.cuda().requires_grad_(True), scene._scale.cuda().requires_grad_(True)
    g = torch.Generator().manual_seed(3)
    raw = (4.0 * torch.randn(s.shape[0], 1, generator=g)).cuda().requires_grad_(True)
    up = torch.randn(s.shape[0], 1, generator=g).cuda()
    out = mesh_to_gaussians(v, f, a, s, "relu", fused_activations=True, _opacity=raw)
    assert len(out) == 7 and out[6].shape == raw.shape
    ref = torch.sigmoid(raw.detach().clone().requires_grad_(True))
    assert float((out[6] - ref).abs().max()) <= 2e-7
    ((out[6] * up).sum() + out[1].sum() + out[4].sum() + out[5].sum()).backward()
    raw2 = raw.detach().clone().requires_grad_(True)
    (torch.sigmoid(raw2) * up).sum().backward()
    assert float((raw.grad - raw2.grad).abs().max()) <= 1e-6 * float(raw2.grad.abs().max()) + 1e-9
    assert torch.isfinite(v.grad).all() and float(v.grad.abs().max()) > 0


def test_model_getter_falls_back_when_opacity_changes_after_update_alpha():
    from games_hip.model import HipGaussianMeshModel
    m = HipGaussianMeshModel.from_scene(syn.mesh_scene("tiny"), "cuda")
    m.update_alpha(); m.prepare_scaling_rot()
    fused = m.get_opacity
    assert float((fused - torch.sigmoid(m._opacity)).abs().max()) <= 2e-7
    with torch.no_grad():
        m._opacity.add_(1.0)                       # e.g. an optimizer step without update_alpha()
    assert float((m.get_opacity - torch.sigmoid(m._opacity)).abs().max()) == 0.0


def test_checkpoint_roundtrip_in_the_reference_file_format(tmp_path):
    """save_ply / load_ply: point_cloud.ply with the reference's attribute list + model_params.pt."""
    from games_hip.model import HipGaussianMeshModel
    from games_hip.render import PipelineParams, render
    from plyfile import PlyData
    m = HipGaussianMeshModel.from_scene(syn.mesh_scene("tiny"), "cuda")
    path = str(tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply")
    m.save_ply(path)
    el = PlyData.read(path).elements[0]
    assert [p.name for p in el.properties][:9] == ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
    assert el.count == m._xyz.shape[0] and np.allclose(np.asarray(el["scale_1"]), m._scaling[:, 1].detach().cpu().numpy())
    m2 = HipGaussianMeshModel(3)
    m2.alpha_mode = m.alpha_mode
    m2.load_ply(path)
    for a in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest"):
        assert torch.equal(getattr(m, a).detach(), getattr(m2, a).detach()), a
    cam = syn.orbit_camera(0, width=96, height=96).to("cuda")
    bg = torch.ones(3, device="cuda")
    assert torch.equal(render(cam, m, PipelineParams(), bg)["render"], render(cam, m2, PipelineParams(), bg)["render"])


def test_backward_twice_through_one_graph_and_without_vertex_gradient():
    """The forward pre-clears the vertex-gradient buffer for ONE backward (fused single launch); a second backward through
    a retained graph takes the two-launch path with its own clearing, and a graph without vertex gradient needs none."""
    from games_hip.mesh_op import mesh_to_gaussians
    scene = syn.mesh_scene("tiny")
    f = scene.faces.cuda()
    v = scene.vertices.cuda().requires_grad_(True)
    a, s = scene._alpha.cuda().requires_grad_(True), scene._scale.cuda().requires_grad_(True)
    out = mesh_to_gaussians(v, f, a, s, "relu", fused_activations=True)
    loss = out[1].sum() + (out[4] * out[4]).sum() + out[5][:, 1].sum()
    loss.backward(retain_graph=True)
    g1 = [t.grad.clone() for t in (v, a, s)]
    for t in (v, a, s):
        t.grad = None
    loss.backward()
    for x, y in zip(g1, (v.grad, a.grad, s.grad)):
        _close(y, x, rtol=1e-5)
    v2 = scene.vertices.cuda()                       # no gradient for the vertices
    a2 = scene._alpha.cuda().requires_grad_(True)
    out2 = mesh_to_gaussians(v2, f, a2, scene._scale.cuda(), "relu", fused_activations=True)
    (out2[1].sum() + (out2[4] * out2[4]).sum() + out2[5][:, 1].sum()).backward()
    _close(a2.grad, g1[1], rtol=1e-5)
