"""CPU, authoring container only: the REFERENCE's own `render()` (renderer/gaussian_renderer/__init__.py:25-111) and its
own `GaussianMeshModel` run unmodified on top of the drop-in package's Python surface -- `GaussianRasterizationSettings`,
`GaussianRasterizer.forward` with the reference's keyword call, the (color, radii, invdepth) return, the autograd contract
with `screenspace_points.grad` -- with only the innermost kernel call replaced by the CPU oracle (there is no GPU here and
no reference tree on the GPU box, so this is the one place where the reference's glue and the drop-in meet in a test).
With `install()`: the mixin model (K0 op replaced by the restatement) renders the same image as the reference class."""
import os

import numpy as np
import pytest
import torch

from games_hip import synthetic as syn
from oracle import gs_oracle, ref_import


class _OracleRaster(torch.autograd.Function):
    """Test-only stand-in for the kernels: same inputs / outputs as `_rasterize_gaussians`, computed by the C oracle."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors, opacities, scales, rotations, cov, rs):
        nz = lambda t: t.detach() if (t is not None and t.numel()) else None
        o = gs_oracle.rasterize(means3D=means3D.detach(), opacities=opacities.detach(), shs=nz(sh), colors_precomp=nz(colors),
                                scales=nz(scales), rotations=nz(rotations), cov3D_precomp=nz(cov),
                                image_height=rs.image_height, image_width=rs.image_width, tanfovx=rs.tanfovx, tanfovy=rs.tanfovy,
                                bg=rs.bg, scale_modifier=rs.scale_modifier, viewmatrix=rs.viewmatrix, projmatrix=rs.projmatrix,
                                sh_degree=rs.sh_degree, campos=rs.campos, antialiasing=rs.antialiasing)
        ctx.o = o
        ctx.has = (sh.numel() > 0, colors.numel() > 0, scales.numel() > 0, cov.numel() > 0)
        radii = torch.from_numpy(o.radii.copy())
        ctx.mark_non_differentiable(radii)
        return torch.from_numpy(o.color.copy()), radii, torch.from_numpy(o.invdepth.copy())

    @staticmethod
    def backward(ctx, g_color, _g_radii, g_invd):
        g = gs_oracle.backward(ctx.o, g_color, g_invd if g_invd is not None and g_invd.abs().sum() > 0 else None)
        t = lambda k: torch.from_numpy(np.asarray(g[k]).copy())
        has_sh, has_col, has_sr, has_cov = ctx.has
        return (t("means3D"), t("means2D"), t("sh") if has_sh else None, t("colors_precomp") if has_col else None, t("opacities"),
                t("scales") if has_sr else None, t("rotations") if has_sr else None, t("cov3D_precomp") if has_cov else None, None)


def _oracle_rasterize(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings, visible_out=None):
    from diff_gaussian_rasterization import SplitSH
    if isinstance(sh, SplitSH):
        sh = sh.full()
    return _OracleRaster.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings)


class _Pipe:
    convert_SHs_python = False
    compute_cov3D_python = False
    debug = False
    antialiasing = False


@pytest.mark.parametrize("installed", [False, True])
def test_reference_render_and_backward_run_on_the_drop_in_surface(monkeypatch, installed):
    if not ref_import.available():
        pytest.skip("reference tree not present")
    ref = ref_import.import_reference()
    import importlib
    import diff_gaussian_rasterization as dgr
    assert not getattr(dgr, "__games_stub__", False)
    monkeypatch.setattr(dgr, "_rasterize_gaussians", _oracle_rasterize)
    import games
    from games_hip import model as hip_model
    import test_abi
    monkeypatch.setattr(hip_model, "mesh_to_gaussians", test_abi._cpu_op)
    ref_render = importlib.import_module("renderer.gaussian_renderer").render          # the reference's render(), unmodified
    out = hip_model.install(games) if installed else {}
    try:
        scene = syn.mesh_scene("tiny")
        with ref_import.cuda_literals_on_cpu():
            m = games.gaussianModel["gs_mesh"](3)
            m.vertices = torch.nn.Parameter(scene.vertices.clone())
            m.faces = scene.faces
            m._alpha = torch.nn.Parameter(scene._alpha.clone())
            m._scale = torch.nn.Parameter(scene._scale.clone())
            m._opacity = torch.nn.Parameter(scene._opacity.clone())
            m._features_dc = torch.nn.Parameter(scene._features_dc.clone())
            m._features_rest = torch.nn.Parameter(scene._features_rest.clone())
            m.active_sh_degree = 3
            m.update_alpha()
            m.prepare_scaling_rot()
            cam = syn.orbit_camera(2, width=64, height=48)
            bg = torch.tensor([1.0, 1.0, 1.0])
            pkg = ref_render(cam, m, _Pipe(), bg)
            image = pkg["render"]
            assert image.shape == (3, 48, 64) and pkg["radii"].dtype == torch.int32 and pkg["depth"].shape == (1, 48, 64)
            assert pkg["visibility_filter"].dtype == torch.bool and int(pkg["visibility_filter"].sum()) > 0
            loss = ((image - 0.5) ** 2).mean()
            loss.backward()
        # gradients reached every model parameter through the reference's getters and the K0 stage, and the screen-space leaf
        for p in (m.vertices, m._alpha, m._scale, m._opacity, m._features_dc, m._features_rest):
            assert p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0
        assert pkg["viewspace_points"].grad is not None and float(pkg["viewspace_points"].grad.abs().max()) > 0
        # the picture equals the oracle rendering of the restatement's Gaussians
        from oracle import mesh_oracle
        with torch.no_grad():
            _, _, xyz, scaling, rot = mesh_oracle.mesh_to_gaussians(scene.vertices, scene.faces, scene._alpha, scene._scale)
            xa, sa, ra, oa, shs = mesh_oracle.activated(xyz, scaling, rot, scene._opacity, scene._features_dc, scene._features_rest)
        o = gs_oracle.rasterize(means3D=xa, opacities=oa, shs=shs, scales=sa, rotations=ra, image_height=48, image_width=64,
                                tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=bg, viewmatrix=cam.world_view_transform,
                                projmatrix=cam.full_proj_transform, sh_degree=3, campos=cam.camera_center)
        assert float(np.abs(image.detach().numpy() - o.color).max()) <= 1e-5
    finally:
        if out:
            hip_model.uninstall(games, out)
        ref_import.drop_reference_stubs()


def test_reference_animated_renderer_with_the_installed_mixin(monkeypatch):
    """renderer/gaussian_animated_renderer/__init__.py:21-121 (scripts/render_time_animated.py drives it): it assigns
    `pc.triangles`, calls `pc.prepare_scaling_rot()` and takes the centres from `pc.alpha @ triangles`; with install() the
    mixin derives scale / rotation from the assigned triangles.  Image == oracle rendering of the deformed mesh."""
    if not ref_import.available():
        pytest.skip("reference tree not present")
    ref_import.import_reference()
    import importlib
    import diff_gaussian_rasterization as dgr
    import games
    from games_hip import model as hip_model
    from oracle import mesh_oracle
    import test_abi
    monkeypatch.setattr(dgr, "_rasterize_gaussians", _oracle_rasterize)
    monkeypatch.setattr(hip_model, "mesh_to_gaussians", test_abi._cpu_op)

    def cpu_tri_op(triangles, _alpha, _scale, alpha_mode="relu", fused_activations=False):
        F = triangles.shape[0]
        return test_abi._cpu_op(triangles.reshape(3 * F, 3), torch.arange(3 * F).reshape(F, 3), _alpha, _scale, alpha_mode,
                                fused_activations=fused_activations)
    monkeypatch.setattr(hip_model, "triangles_to_gaussians", cpu_tri_op)
    anim_render = importlib.import_module("renderer.gaussian_animated_renderer").render
    out = hip_model.install(games)
    try:
        scene = syn.mesh_scene("tiny")
        new_v = scene.vertices * torch.tensor([1.1, 0.9, 1.0]) + torch.tensor([0.0, 0.03, 0.0])
        with ref_import.cuda_literals_on_cpu(), torch.no_grad():
            m = games.gaussianModelRender["gs_mesh"](3)
            m.vertices, m.faces = scene.vertices.clone(), scene.faces
            m._alpha, m._scale, m._opacity = scene._alpha.clone(), scene._scale.clone(), scene._opacity.clone()
            m._features_dc, m._features_rest = scene._features_dc.clone(), scene._features_rest.clone()
            m.active_sh_degree = 3
            m.update_alpha()
            cam = syn.orbit_camera(5, width=64, height=64)
            bg = torch.ones(3)
            img = anim_render(None, new_v[scene.faces], cam, m, _Pipe(), bg)["render"]
            _, _, xyz, scaling, rot = mesh_oracle.mesh_to_gaussians(new_v, scene.faces, scene._alpha, scene._scale)
            xa, sa, ra, oa, shs = mesh_oracle.activated(xyz, scaling, rot, scene._opacity, scene._features_dc, scene._features_rest)
        o = gs_oracle.rasterize(means3D=xa, opacities=oa, shs=shs, scales=sa, rotations=ra, image_height=64, image_width=64,
                                tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=bg, viewmatrix=cam.world_view_transform,
                                projmatrix=cam.full_proj_transform, sh_degree=3, campos=cam.camera_center)
        assert float(np.abs(img.numpy() - o.color).max()) <= 1e-5
    finally:
        hip_model.uninstall(games, out)
        ref_import.drop_reference_stubs()


def test_reference_points_animated_renderer_runs_on_the_drop_in_surface(monkeypatch):
    """renderer/gaussian_points_animated_renderer/__init__.py:21-114 (scripts/render_points_time_animated.py,
    scripts/render_from_object.py), the fourth `render()` variant of SURVEY 8(a) a7, with the reference's own
    `PointsGaussianModel` (games/flat_splatting/scene/points_gaussian_model.py: pseudo-mesh faces from flat Gaussians, scale /
    rotation re-derived in PyTorch from the deformed faces): centres = first triangle vertex, flat first scale axis 1e-8.
    Image == the oracle rendering of exactly those Gaussians; gradients reach the model."""
    if not ref_import.available():
        pytest.skip("reference tree not present")
    ref_import.import_reference()
    import importlib
    import diff_gaussian_rasterization as dgr
    import games
    monkeypatch.setattr(dgr, "_rasterize_gaussians", _oracle_rasterize)
    pts_render = importlib.import_module("renderer.gaussian_points_animated_renderer").render
    try:
        sc = syn.flat_scene(400, seed=3)
        with ref_import.cuda_literals_on_cpu():
            m = games.gaussianModelRender["gs_points"](3)
            m._xyz = torch.nn.Parameter(sc.means3D.clone())
            m._scaling = torch.nn.Parameter(torch.log(sc.scales[:, 1:].clone()))          # the model keeps the two in-plane axes
            m._rotation = torch.nn.Parameter(sc.rotations.clone())
            m._opacity = torch.nn.Parameter(torch.logit(sc.opacities.clone()))
            m._features_dc = torch.nn.Parameter(sc.shs[:, :1].clone())
            m._features_rest = torch.nn.Parameter(sc.shs[:, 1:].clone())
            m.active_sh_degree = 3
            with torch.no_grad():
                m.prepare_vertices()                                                   # pseudo-mesh faces of the flat Gaussians
            tri = (m.triangles * torch.tensor([1.0, 1.05, 0.95]) + torch.tensor([0.02, 0.0, -0.01])).detach().requires_grad_(True)
            cam = syn.orbit_camera(3, width=72, height=56)
            bg = torch.tensor([0.0, 0.0, 0.0])
            pkg = pts_render(tri, cam, m, _Pipe(), bg)
            image = pkg["render"]
            assert image.shape == (3, 56, 72) and pkg["radii"].dtype == torch.int32
            ((image - 0.3) ** 2).mean().backward()
            assert tri.grad is not None and float(tri.grad.abs().max()) > 0          # through prepare_scaling_rot(triangles) and the centres
            assert m._opacity.grad is not None and m._features_dc.grad is not None
            with torch.no_grad():
                xa, sa, ra, oa = tri[:, 0].detach(), m.get_scaling, m.get_rotation, m.get_opacity
                shs = m.get_features
        o = gs_oracle.rasterize(means3D=xa, opacities=oa, shs=shs, scales=sa, rotations=ra, image_height=56, image_width=72,
                                tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=bg, viewmatrix=cam.world_view_transform,
                                projmatrix=cam.full_proj_transform, sh_degree=3, campos=cam.camera_center)
        assert float(np.abs(image.detach().numpy() - o.color).max()) <= 1e-5 and int((o.radii > 0).sum()) > 50
    finally:
        ref_import.drop_reference_stubs()


def test_reference_flame_renderer_with_the_installed_mixin(monkeypatch):
    """renderer/flame_gaussian_renderer/__init__.py:20-116 (scripts/render_flame.py drives it), the third `render()` variant: the
    centres follow `pc.alpha @ vertices[pc.faces]` for the vertices the caller passes, scale / rotation stay what the model holds
    (SURVEY appendix C.1).  `pc` is the reference's GaussianFlameModel with the installed HIP mixin (K0 op served by the
    restatement here); image == the oracle rendering of those Gaussians."""
    if not ref_import.available():
        pytest.skip("reference tree not present")
    ref_import.import_reference()
    import importlib
    import diff_gaussian_rasterization as dgr
    import games
    from games_hip import model as hip_model
    from oracle import mesh_oracle
    import test_abi
    monkeypatch.setattr(dgr, "_rasterize_gaussians", _oracle_rasterize)
    monkeypatch.setattr(hip_model, "mesh_to_gaussians", test_abi._cpu_op)
    flame_render = importlib.import_module("renderer.flame_gaussian_renderer").flame_render
    out = hip_model.install(games)
    try:
        scene = syn.mesh_scene("tiny")
        new_v = scene.vertices * torch.tensor([0.95, 1.1, 1.0]) + torch.tensor([0.01, 0.0, 0.02])
        with ref_import.cuda_literals_on_cpu(), torch.no_grad():
            m = games.gaussianModelRender["gs_flame"](3)
            par = lambda t: torch.nn.Parameter(t.clone())
            m.point_cloud = hip_model._FlameCloud(hip_model._SyntheticFlameLayer(scene.vertices.clone()), hip_model._squeeze_and_enlarge)
            m.faces = scene.faces
            m._flame_shape, m._flame_exp, m._flame_pose = par(torch.zeros(1, 4)), par(torch.zeros(1, 4)), par(torch.zeros(1, 6))
            m._flame_neck_pose, m._flame_trans = par(torch.zeros(1, 3)), par(torch.zeros(1, 3))
            m._vertices_enlargement = par(torch.ones_like(scene.vertices))
            m._alpha, m._scales, m._opacity = par(scene._alpha), par(scene._scale), par(scene._opacity)
            m._features_dc, m._features_rest = par(scene._features_dc), par(scene._features_rest)
            m.active_sh_degree = 3
            m.update_alpha()                    # FLAME layer (synthetic here) -> vertices -> softmax alphas, xyz  (the mixin)
            m.prepare_scaling_rot()
            cam = syn.orbit_camera(6, width=64, height=64)
            bg = torch.ones(3)
            img = flame_render(cam, m, _Pipe(), bg, vertices=new_v)["render"]
            # what the renderer fed the rasterizer: centres from the NEW vertices, scale / rotation from the model (rest pose)
            al, _, _, scaling, rot = mesh_oracle.mesh_to_gaussians(scene.vertices, scene.faces, scene._alpha, scene._scale, "softmax")
            xyz = torch.matmul(al, new_v[scene.faces]).reshape(-1, 3)
            xa, sa, ra, oa, shs = mesh_oracle.activated(xyz, scaling, rot, scene._opacity, scene._features_dc, scene._features_rest)
        o = gs_oracle.rasterize(means3D=xa, opacities=oa, shs=shs, scales=sa, rotations=ra, image_height=64, image_width=64,
                                tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=bg, viewmatrix=cam.world_view_transform,
                                projmatrix=cam.full_proj_transform, sh_degree=3, campos=cam.camera_center)
        assert float(np.abs(img.numpy() - o.color).max()) <= 1e-5
    finally:
        hip_model.uninstall(games, out)
        ref_import.drop_reference_stubs()


def test_reference_render_time_animated_loop_equals_the_driver(monkeypatch, tmp_path):
    """scripts/render_time_animated.py:68-87: the reference's own `render_set` loop (its `transform_hotdog_fly`, its animated
    renderer, its `torchvision.utils.save_image` calls) runs UNMODIFIED on the drop-in with the installed mesh mixin, and
    `games_hip.animate.render_time_animated` -- the driver the GPU box uses -- produces the same frames bit for bit."""
    if not ref_import.available():
        pytest.skip("reference tree not present")
    ref_import.import_reference()
    import importlib
    import sys
    import types
    import diff_gaussian_rasterization as dgr
    import games
    from games_hip import animate, model as hip_model
    from games_hip.render import PipelineParams
    import test_abi
    monkeypatch.setattr(dgr, "_rasterize_gaussians", _oracle_rasterize)
    monkeypatch.setattr(hip_model, "mesh_to_gaussians", test_abi._cpu_op)

    def cpu_tri_op(triangles, _alpha, _scale, alpha_mode="relu", fused_activations=False):
        F = triangles.shape[0]
        return test_abi._cpu_op(triangles.reshape(3 * F, 3), torch.arange(3 * F).reshape(F, 3), _alpha, _scale, alpha_mode,
                                fused_activations=fused_activations)
    monkeypatch.setattr(hip_model, "triangles_to_gaussians", cpu_tri_op)
    import games_hip.render as hip_render_mod
    monkeypatch.setattr(hip_render_mod, "triangles_to_gaussians", cpu_tri_op, raising=False)
    import games_hip.mesh_op as mesh_op_mod
    monkeypatch.setattr(mesh_op_mod, "triangles_to_gaussians", cpu_tri_op)
    saved = []
    tv = types.ModuleType("torchvision")
    tv.utils = types.SimpleNamespace(save_image=lambda t, path: saved.append((os.path.basename(os.path.dirname(path)), t.detach().clone())))
    monkeypatch.setitem(sys.modules, "torchvision", tv)
    script = importlib.import_module("scripts.render_time_animated")          # the reference's script module (its __main__ part is guarded)
    out = hip_model.install(games)
    try:
        scene = syn.mesh_scene("tiny")
        with ref_import.cuda_literals_on_cpu(), torch.no_grad():
            m = games.gaussianModelRender["gs_mesh"](3)
            m.vertices, m.faces = scene.vertices.clone(), scene.faces
            m._alpha, m._scale, m._opacity = scene._alpha.clone(), scene._scale.clone(), scene._opacity.clone()
            m._features_dc, m._features_rest = scene._features_dc.clone(), scene._features_rest.clone()
            m.active_sh_degree = 3
            m.update_alpha(); m.prepare_scaling_rot()
            views = []
            for k in range(3):
                c = syn.orbit_camera(k, width=48, height=40)
                c.original_image = torch.zeros(3, 40, 48)
                views.append(c)
            bg = torch.ones(3)
            script.render_set(None, str(tmp_path), "test", 7, views, m, _Pipe(), bg)          # the reference's loop, unmodified
            ref_frames = [t for where, t in saved if where == "time_animated"]
            assert len(ref_frames) == 3 and os.path.isdir(tmp_path / "test" / "ours_7" / "time_animated")
            ours = animate.render_time_animated(m, views, PipelineParams(), bg, transform=animate.transform_hotdog_fly)
        for a, b in zip(ref_frames, ours):
            assert torch.equal(a, b)
        assert not torch.equal(ours[0], ours[2])                      # the frames differ: the deformation was applied
    finally:
        hip_model.uninstall(games, out)
        ref_import.drop_reference_stubs()
        sys.modules.pop("scripts.render_time_animated", None)
