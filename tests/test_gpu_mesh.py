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
    """Forward AND backward against the reference-executed fixture.  k0_mesh.npz carries the edge cases of
    gaussian_mesh_model.py:103-169: relu-clipped alpha rows, a negative `_scale` entry (all three scales collapse to eps,
    zero gradient) and two degenerate faces (repeated vertex, point face).  Degenerate faces: the forward must give the
    reference's own finite values; their gradients are compared everywhere except on the degenerate faces' own vertices /
    splats, where the reference differentiates 0/0-type norms by torch's sub-gradient convention."""
    from games_hip.mesh_op import mesh_to_gaussians
    g = _load(golden_dir, name)
    v = g["vertices"].cuda().requires_grad_(True)
    a = g["_alpha"].cuda().requires_grad_(True)
    s = g["_scale"].cuda().requires_grad_(True)
    alpha, xyz, scaling, rot = mesh_to_gaussians(v, g["faces"].cuda(), a, s, "relu")
    _close(alpha, g["alpha"], rtol=1e-6)
    _close(xyz, g["xyz"], rtol=1e-5)
    tri = g["triangles"]
    area = torch.linalg.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]).norm(dim=1)
    S = g["_alpha"].shape[1]
    okf = area > 1e-9                                   # non-degenerate faces
    ok = okf.repeat_interleave(S)
    _close(scaling[ok.cuda()], g["scaling"][ok], rtol=1e-5)
    _close(rot[ok.cuda()], g["rotation"][ok], rtol=1e-4)
    assert torch.isfinite(rot).all() and torch.isfinite(scaling).all()
    if not ok.all():
        # degenerate faces: "the same finite values" as the reference.  log-scales sit at
        # log(eps)-like values (-17..-18.4) where one float ulp of s*eps moves the log by ~1e-7 relative: abs tolerance
        bad = ~ok
        assert float((scaling[bad.cuda()].cpu() - g["scaling"][bad]).abs().max()) < 2e-2
        _close(torch.exp(scaling[bad.cuda()]), torch.exp(g["scaling"][bad]), rtol=1e-3, atol_rel=1e-3)
        point = (tri[:, 0] == tri[:, 1]).all(1) & (tri[:, 1] == tri[:, 2]).all(1)         # all three vertices equal
        pt = point.repeat_interleave(S)
        _close(rot[pt.cuda()], g["rotation"][pt], rtol=1e-5)        # zero frame -> quaternion (0.5, 0, 0, 0)
    # negative _scale entries: relu(_scale * s) = 0 -> every log-scale = log(eps) exactly as the reference
    neg = (g["_scale"][:, 0] < 0)
    if neg.any():
        assert torch.equal(scaling[neg.cuda()].cpu(), g["scaling"][neg])
    _loss(xyz, scaling, rot, g, "cuda").backward()
    assert torch.isfinite(v.grad).all() and torch.isfinite(a.grad).all() and torch.isfinite(s.grad).all()
    touched = torch.zeros(g["vertices"].shape[0], dtype=torch.bool)
    touched[g["faces"][~okf].reshape(-1)] = True        # vertices of degenerate faces
    _close(v.grad.cpu()[~touched], g["d_vertices"][~touched])
    _close(a.grad.cpu()[okf], g["d_alpha"][okf])
    _close(s.grad.cpu()[ok], g["d_scale"][ok])
    if neg.any():
        assert float(s.grad.cpu()[neg].abs().max()) == 0.0 and float(g["d_scale"][neg].abs().max()) == 0.0
    # d xyz / d alpha does not involve the frame: exact on degenerate faces too
    if not okf.all():
        _close(a.grad.cpu()[~okf], g["d_alpha"][~okf])


def test_negative_and_zero_scale_backward_against_restatement():
    """`_scale <= 0` rows (relu closes the gate: log(eps) forward, zero gradient) mixed with ordinary ones, both backward
    paths (thread-per-face S=3, wave-per-face S=20)."""
    from games_hip.mesh_op import mesh_to_gaussians
    for S in (3, 20):
        v, f = syn.uv_sphere(7, 9)
        gen = torch.Generator().manual_seed(50 + S)
        F = f.shape[0]
        a = torch.rand(F, S, 3, generator=gen)
        s = torch.exp(0.3 * torch.randn(F * S, 1, generator=gen))
        s[::5] = -s[::5]
        s[3::11] = 0.0
        g = dict(g_xyz=torch.randn(F * S, 3, generator=gen), g_scaling_act=torch.randn(F * S, 3, generator=gen),
                 g_rotation_act=torch.randn(F * S, 4, generator=gen))
        vc, ac, sc = v.clone().requires_grad_(True), a.clone().requires_grad_(True), s.clone().requires_grad_(True)
        _, _, xyz, scaling, rot = mesh_oracle.mesh_to_gaussians(vc, f, ac, sc, "relu")
        _loss(xyz, scaling, rot, g, "cpu").backward()
        vg, ag, sg = v.cuda().requires_grad_(True), a.cuda().requires_grad_(True), s.cuda().requires_grad_(True)
        _, xyz_h, scaling_h, rot_h = mesh_to_gaussians(vg, f.cuda(), ag, sg, "relu")
        _loss(xyz_h, scaling_h, rot_h, g, "cuda").backward()
        closed = (s[:, 0] <= 0)
        assert torch.equal(scaling_h.cpu()[closed], scaling.detach()[closed])
        assert float(sg.grad.cpu()[closed].abs().max()) == 0.0 and float(sc.grad[closed].abs().max()) == 0.0
        _close(scaling_h, scaling, rtol=1e-5); _close(vg.grad, vc.grad); _close(ag.grad, ac.grad); _close(sg.grad, sc.grad)


def test_flame_mixin_matches_reference_fixture(golden_dir):
    """HipFlameMixin on a host with the attributes of GaussianFlameModel (gaussian_flame_model.py:27-50): softmax alpha,
    vertices out of the FLAME layer + transform_vertices_function, `_scales`; fixture = the reference class executed."""
    from types import SimpleNamespace
    from games_hip.model import HipFlameMixin
    g = _load(golden_dir, "k0_flame.npz")

    class Host(HipFlameMixin):
        pass

    def transform(vertices, c):                       # games/flame_splatting/scene/dataset_readers.py:41-46
        vv = torch.squeeze(vertices)
        return torch.stack([vv[:, 0], -vv[:, 2], vv[:, 1]], dim=1) * c
    m = Host()
    v0 = g["flame_vertices"].cuda().requires_grad_(True)
    m.point_cloud = SimpleNamespace(flame_model=lambda **kw: (v0[None], None), transform_vertices_function=transform)
    m.faces = g["faces"].cuda()
    z = lambda *sh: torch.zeros(*sh, device="cuda")
    m._flame_shape, m._flame_exp, m._flame_pose, m._flame_neck_pose, m._flame_trans = z(1, 4), z(1, 4), z(1, 6), z(1, 3), z(1, 3)
    m._vertices_enlargement = g["enlargement"].cuda().requires_grad_(True)
    m._alpha = g["_alpha"].cuda().requires_grad_(True)
    m._scales = torch.empty(0)
    m.update_alpha()                                   # create_from_pcd order: before `_scales` exists (:78-82)
    _close(m.alpha, g["alpha"], rtol=1e-5); _close(m._xyz, g["xyz"], rtol=1e-5); _close(m.vertices, g["vertices"], rtol=1e-6)
    m._scales = g["_scales"].cuda().requires_grad_(True)
    m._opacity = torch.zeros(g["_scales"].shape[0], 1, device="cuda")
    m.prepare_scaling_rot()
    _close(m._scaling, g["scaling"], rtol=1e-5); _close(m._rotation, g["rotation"], rtol=1e-4)
    m.update_alpha(); m.prepare_scaling_rot()          # the per-iteration order of train.py:154-157
    _close(m._scaling, g["scaling"], rtol=1e-5); _close(m._rotation, g["rotation"], rtol=1e-4)
    assert float((m.get_opacity - 0.5).abs().max()) < 1e-6
    ((m.get_xyz if hasattr(m, "get_xyz") else m._xyz) * g["g_xyz"].cuda()).sum().backward(retain_graph=True)
    ((m.get_scaling * g["g_scaling_act"].cuda()).sum() + (m.get_rotation * g["g_rotation_act"].cuda()).sum()).backward()
    _close(v0.grad, g["d_flame_vertices"]); _close(m._vertices_enlargement.grad, g["d_enlargement"])
    _close(m._alpha.grad, g["d_alpha"]); _close(m._scales.grad, g["d_scales"])


def test_standalone_multi_mesh_and_flame_models_render_and_train_step():
    """The stand-alone gs_multi_mesh / gs_flame hosts used by bench.py: one fwd+bwd through render() gives finite
    gradients on every parameter, and the model-level outputs equal the restatement's."""
    from games_hip.model import HipGaussianFlameModel, HipGaussianMultiMeshModel
    from games_hip.render import PipelineParams, render
    scenes = syn.multi_mesh_scenes("multi_tiny")
    mm = HipGaussianMultiMeshModel.from_scenes(scenes, "cuda")
    xyz_o, scaling_o, rot_o = mesh_oracle.multi_mesh_to_gaussians([s.vertices for s in scenes], [s.faces for s in scenes],
                                                                  [s._alpha for s in scenes], [s._scale for s in scenes])
    _close(mm.get_xyz, xyz_o, rtol=1e-5); _close(mm._scaling, scaling_o, rtol=1e-5); _close(mm._rotation, rot_o, rtol=1e-4)
    fm = HipGaussianFlameModel.from_scene(syn.mesh_scene("tiny"), "cuda")
    cam = syn.orbit_camera(1, width=64, height=64).to("cuda")
    bg = torch.ones(3, device="cuda")
    for model in (mm, fm):
        model.update_alpha(); model.prepare_scaling_rot()
        img = render(cam, model, PipelineParams(), bg)["render"]
        assert torch.isfinite(img).all() and float((img - 1.0).abs().max()) > 0.05       # something was drawn
        img.backward((img.detach() - 0.5) / img.numel())
        for p in model.parameters():
            assert p.grad is not None and torch.isfinite(p.grad).all()
    assert float(fm._flame_exp.grad.abs().max()) > 0 and float(mm.vertices[1].grad.abs().max()) > 0


def test_reference_call_order_save_dict_and_cache_invalidation(tmp_path):
    """What the reference's own methods do to a model carrying HipMeshMixin: create_from_pcd calls
    update_alpha() while `_scale` is still torch.empty(0) (gaussian_mesh_model.py:78-81); save_ply reads `triangles` out
    of the instance __dict__ (:193-207); prepare_scaling_rot() alone after editing vertices / _scale must not serve the
    values cached by the last update_alpha(); FusedAdam.step() must invalidate the fused opacity getter."""
    from games_hip.model import HipMeshMixin
    from games_hip.optim import FusedAdam
    scene = syn.mesh_scene("tiny")

    class Base:                                        # the slice of GaussianMeshModel.save_ply that matters here
        def save_ply(self, path):
            self.update_alpha(); self.prepare_scaling_rot()
            attrs = self.__dict__
            torch.save({k: attrs[k] for k in ("_alpha", "_scale", "triangles", "vertices", "faces")}, path)

    class Host(HipMeshMixin, Base):
        pass
    m = Host()
    m.triangles = None                                 # GaussianMeshModel.__init__ :47
    m.vertices = torch.nn.Parameter(scene.vertices.cuda())
    m.faces = scene.faces.cuda()
    m._alpha = torch.nn.Parameter(scene._alpha.cuda())
    m._scale = torch.empty(0)
    m.update_alpha()                                   # :78, before _scale is set
    _, _, xyz_o, scaling_o, rot_o = mesh_oracle.mesh_to_gaussians(scene.vertices, scene.faces, scene._alpha, scene._scale)
    _close(m._xyz, xyz_o, rtol=1e-5)
    m._scale = torch.nn.Parameter(scene._scale.cuda())
    m.prepare_scaling_rot()                            # :82
    m._opacity = torch.nn.Parameter(scene._opacity.cuda())
    _close(m._scaling, scaling_o, rtol=1e-5); _close(m._rotation, rot_o, rtol=1e-4)
    path = str(tmp_path / "model_params.pt")
    m.save_ply(path)
    saved = torch.load(path, weights_only=False)
    assert torch.equal(saved["triangles"].cpu(), scene.vertices[scene.faces])
    # stale-cache check: edit vertices and _scale in place, call ONLY prepare_scaling_rot()
    with torch.no_grad():
        m.vertices.mul_(torch.tensor([1.0, 1.4, 0.7], device="cuda"))
        m._scale.mul_(1.3)
    m.prepare_scaling_rot()
    v2 = scene.vertices * torch.tensor([1.0, 1.4, 0.7])
    _, _, _, scaling2, rot2 = mesh_oracle.mesh_to_gaussians(v2, scene.faces, scene._alpha, scene._scale * 1.3)
    _close(m._scaling, scaling2, rtol=1e-5); _close(m._rotation, rot2, rtol=1e-4)
    # FusedAdam writes through raw pointers: the fused get_opacity must notice
    m.update_alpha(); m.prepare_scaling_rot()
    before = m.get_opacity.clone()
    opt = FusedAdam([{"params": [m._opacity], "lr": 0.5}], lr=0.0, eps=1e-15)
    m._opacity.grad = torch.ones_like(m._opacity)
    opt.step()
    after = m.get_opacity
    assert float((after - torch.sigmoid(m._opacity)).abs().max()) <= 2e-7 and float((after - before).abs().max()) > 0.05


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
    from games_hip._plyfile_compat import PlyData
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


def test_flame_checkpoint_roundtrip_renders_identically(tmp_path):
    """flame_params.pt + point_cloud.ply (gaussian_flame_model.py:232-265) through the HIP op: the loaded model renders the
    same image and keeps animating with re-derived face-local scale / rotation."""
    from games_hip.model import HipGaussianFlameModel
    from games_hip.render import PipelineParams, render
    m = HipGaussianFlameModel.from_scene(syn.mesh_scene("tiny"), "cuda")
    with torch.no_grad():
        m._flame_exp.fill_(0.4); m._flame_pose[0, 0] = 0.3
    path = str(tmp_path / "point_cloud" / "iteration_9" / "point_cloud.ply")
    m.save_ply(path)
    m2 = HipGaussianFlameModel(3)
    m2.load_ply(path)
    cam = syn.orbit_camera(1, width=96, height=96).to("cuda")
    bg = torch.ones(3, device="cuda")
    assert torch.equal(render(cam, m, PipelineParams(), bg)["render"], render(cam, m2, PipelineParams(), bg)["render"])
    with torch.no_grad():
        m._flame_exp.fill_(-0.5); m2._flame_exp.fill_(-0.5)
    for g in (m, m2):
        g.update_alpha(); g.prepare_scaling_rot()
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


def test_animated_drivers_follow_the_reference_loops(tmp_path):
    """render_time_animated (scripts/render_time_animated.py:68-87) and the FLAME driver (scripts/render_flame.py:29-60
    shape): frames follow the deformation, frame k equals a manual render of the k-th deformed mesh, PNGs are written."""
    from games_hip import animate
    from games_hip.model import HipGaussianFlameModel, HipGaussianMeshModel
    from games_hip.render import PipelineParams, render_animated
    scene = syn.mesh_scene("small")
    model = HipGaussianMeshModel.from_scene(scene, "cuda")
    views = [syn.orbit_camera(k, width=96, height=96).to("cuda") for k in range(4)]
    bg = torch.ones(3, device="cuda")
    frames = animate.render_time_animated(model, views, PipelineParams(), bg, transform=animate.transform_ship_sinus,
                                          out_dir=str(tmp_path / "anim"))
    assert len(frames) == 4 and sorted(os.listdir(tmp_path / "anim")) == [f"{k:05d}.png" for k in range(4)]
    ts = torch.linspace(0, 10 * torch.pi, 4)
    rest = model.vertices.detach().clone()
    with torch.no_grad():
        # the reference's ship / ficus transforms write in place (scripts/render_time_animated.py:52-55): frame k shows the
        # deformation ACCUMULATED over frames 0..k
        v2 = rest.clone()
        for k in range(3):
            v2 = animate.transform_ship_sinus(v2, ts[k])
        ref = render_animated(None, v2[model.faces].float(), views[2], model, PipelineParams(), bg)["render"]
    assert torch.equal(frames[2], ref) and not torch.equal(frames[2], frames[1])
    assert torch.equal(model.vertices.detach(), rest)           # the model's own vertices are left alone
    # the copy-returning transforms restart from the rest pose each frame (:35-41)
    frames_fly = animate.render_time_animated(model, views, PipelineParams(), bg, transform=animate.transform_hotdog_fly)
    with torch.no_grad():
        ref_fly = render_animated(None, animate.transform_hotdog_fly(rest, ts[3])[model.faces].float(), views[3], model, PipelineParams(), bg)["render"]
    assert torch.equal(frames_fly[3], ref_fly)
    # both sinus terms of the ficus transform (:28-32)
    z0 = rest[:, 2].clone()
    vf = animate.transform_ficus_sinus(rest.clone(), ts[1], None)
    want = z0 + 0.005 * torch.sin(rest[:, 0] * 2 * torch.pi + ts[1]) + 0.005 * torch.sin(rest[:, 1] * 5 * torch.pi + ts[1])
    assert float((vf[:, 2] - want).abs().max()) <= 1e-6
    fm = HipGaussianFlameModel.from_scene(syn.mesh_scene("tiny"), "cuda")

    def drive(g, k):
        with torch.no_grad():
            g._flame_exp.fill_(0.5 * k)
            g._flame_pose[0, 0] = 0.2 * k
    fr = animate.render_flame_animated(fm, views[:3], PipelineParams(), bg, drive)
    assert len(fr) == 3 and all(torch.isfinite(f).all() for f in fr)
    # same camera, different expression / pose -> different picture; frame 0 = the undeformed model
    fm2 = HipGaussianFlameModel.from_scene(syn.mesh_scene("tiny"), "cuda")
    a = animate.render_flame_animated(fm2, [views[0]], PipelineParams(), bg, lambda g, k: None)[0]
    b = animate.render_flame_animated(fm2, [views[0]], PipelineParams(), bg, lambda g, k: g._flame_exp.data.fill_(1.0))[0]
    assert torch.equal(fr[0], a) and float((a - b).abs().max()) > 1e-3
