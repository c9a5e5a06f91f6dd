"""GPU parity tests of the HIP rasterizer (through the python drop-in -> C ABI -> kernels) against
the CPU oracle.  Tolerances are BASELINE.json's: rendered RGB <= 1e-4 abs, gradients <= 1e-3 rel.

Discontinuity rule (DESIGN.md "Parity"): the rasterizer contains hard thresholds (alpha >= 1/255,
T < 1e-4, ceil() radius, tile rectangle).  The oracle flags every pixel / Gaussian whose decision
was within float rounding of a threshold; those few are compared with the loose bound a flipped
decision implies instead of 1e-4, and their fraction is asserted to be tiny."""
import numpy as np
import pytest
import torch

import _util as U
from games_hip import synthetic as syn

pytestmark = pytest.mark.gpu

RGB_TOL = 1e-4
GRAD_REL = 1e-3


def _check(inputs, kw, W, H, gd=True, q=0.999):
    o = U.oracle_render(inputs, kw)
    gc = syn.upstream_grad(torch.from_numpy(o["color"])).numpy() * 1000.0
    gdm = np.full((1, H, W), 1e-3, np.float32) if gd else None
    o = U.oracle_render(inputs, kw, gc, gdm)
    # gradients: deterministic mode under the strict criterion first, then the default (float atomics) mode under the default one
    h, g = U.assert_grads_both_modes(lambda: U.hip_render(inputs, kw, grad_color=gc, grad_invdepth=gdm), o["grads"],
                                     lambda: U.oracle_render(inputs, kw, gc, gdm, precision="f64")["grads"], q=q,
                                     where=f"P={inputs['means3D'].shape[0]} {W}x{H}", excuse=U.excused_rows(o["details"]),
                                     go32acc_fn=lambda: U.f32_realisations(inputs, kw, gc, gdm),
                                     alt=U.alt_oracles(inputs, kw, gc, gdm, o["details"]), small=True)
    rep = U.forward_report(h, o, W, H)
    assert rep["radii_unexplained"] == 0, rep
    assert rep["amb_frac"] < 0.01, rep
    assert rep["max_clean"] <= RGB_TOL, rep
    assert rep["max_invdepth_clean"] <= RGB_TOL, rep
    assert rep["max_amb"] <= 0.02, rep          # a flipped alpha>=1/255 decision moves a pixel by < 1/255 * max colour
    return h, o, rep, g


def _inputs(sc):
    return dict(means3D=sc.means3D, opacities=sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)


@pytest.mark.parametrize("case", ["random", "aa_deg2", "deg0", "flat10k", "odd_size", "scale_mod"])
def test_forward_backward_parity(case):
    bg = torch.tensor([0.2, 0.4, 0.6])
    if case == "random":
        sc, cam, extra = syn.random_scene(4000, seed=1, scale_lo=0.01, scale_hi=0.1), syn.orbit_camera(1, width=160, height=128, radius=3.0), {}
    elif case == "aa_deg2":
        sc, cam, extra = syn.random_scene(3000, seed=2, scale_lo=0.005, scale_hi=0.08), syn.orbit_camera(2, width=200, height=120, radius=3.0), dict(antialiasing=True, sh_degree=2)
    elif case == "deg0":
        sc, cam, extra = syn.random_scene(2000, seed=3), syn.orbit_camera(3, width=96, height=96, radius=3.5), dict(sh_degree=0)
    elif case == "flat10k":     # BASELINE config 1 inputs (gs_flat: first scale axis 1e-8)
        sc, cam, extra = syn.flat_scene(10000), syn.orbit_camera(0, width=256, height=256), {}
        bg = torch.ones(3)
    elif case == "odd_size":    # width/height not multiples of 16: partial edge tiles
        sc, cam, extra = syn.random_scene(2500, seed=5, scale_lo=0.02, scale_hi=0.2), syn.orbit_camera(5, width=131, height=77, radius=3.0), {}
    else:
        sc, cam, extra = syn.random_scene(2000, seed=6), syn.orbit_camera(6, width=128, height=128, radius=3.0), dict(scale_modifier=1.7)
    kw = U.settings_kwargs(cam, bg, **extra)
    _check(_inputs(sc), kw, cam.image_width, cam.image_height)


def test_upstream_scale_modifier_quirk_switch():
    """include/gmsplat.h, gms_set_upstream_scale_mod_grad: with the switch on dL/dscale loses the scale_modifier factor (what the
    public upstream backward is believed to return) and nothing else changes; at scale_modifier = 1 -- every render() of the
    reference -- the two settings agree."""
    import diff_gaussian_rasterization as dgr
    sc, cam = syn.random_scene(2000, seed=6), syn.orbit_camera(6, width=128, height=128, radius=3.0)
    inputs = _inputs(sc)
    try:
        for mod in (1.7, 1.0):
            kw = U.settings_kwargs(cam, torch.tensor([0.2, 0.4, 0.6]), scale_modifier=mod)
            gc = syn.upstream_grad(torch.from_numpy(U.hip_render(inputs, kw, need_grad=False)["color"])).numpy() * 1000.0
            dgr.set_upstream_scale_mod_grad(False)
            off = U.hip_render(inputs, kw, grad_color=gc)["grads"]
            dgr.set_upstream_scale_mod_grad(True)
            on = U.hip_render(inputs, kw, grad_color=gc)["grads"]
            for k in ("means3D", "means2D", "shs", "opacities", "rotations", "scales"):
                want = off[k] / mod if k == "scales" else off[k]
                tol = 2e-5 * np.abs(want).max() + 1e-12          # two runs differ by the order of the float atomics
                assert np.abs(on[k] - want).max() <= tol, (mod, k, np.abs(on[k] - want).max(), tol)
            assert np.abs(off["scales"]).max() > 0
    finally:
        dgr.set_upstream_scale_mod_grad(False)


def test_precomputed_colors_and_cov3d_inputs():
    from oracle import dense_torch
    sc = syn.random_scene(3000, seed=7, scale_lo=0.01, scale_hi=0.12)
    cam = syn.orbit_camera(4, width=144, height=112, radius=3.0)
    cov = dense_torch.cov3d_python(sc.scales, 1.0, sc.rotations)
    cols = torch.rand(3000, 3, generator=torch.Generator().manual_seed(0))
    inputs = dict(means3D=sc.means3D, opacities=sc.opacities, colors_precomp=cols, cov3D_precomp=cov)
    _check(inputs, U.settings_kwargs(cam, torch.zeros(3), sh_degree=0), 144, 112)


def test_python_stage_flags_give_the_same_image():
    """convert_SHs_python / compute_cov3D_python (renderer/gaussian_renderer/__init__.py:71-91) must not
    change the picture: the rasterizer's native SH / cov3D stages equal the reference's python stages."""
    from games_hip.model import HipGaussianMeshModel
    from games_hip.render import PipelineParams, render
    model = HipGaussianMeshModel.from_scene(syn.mesh_scene("small"), "cuda")
    cam = syn.orbit_camera(1, width=128, height=128).to("cuda")
    bg = torch.ones(3, device="cuda")
    with torch.no_grad():
        base = render(cam, model, PipelineParams(), bg)["render"]
        py_sh = render(cam, model, PipelineParams(convert_SHs_python=True), bg)["render"]
        py_cov = render(cam, model, PipelineParams(compute_cov3D_python=True), bg)["render"]
    assert (base - py_sh).abs().max().item() <= 2e-5
    assert (base - py_cov).abs().max().item() <= 2e-5
    assert base.std().item() > 0.01


def test_empty_culled_and_argument_errors():
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    cam = syn.orbit_camera(0, width=48, height=32).to("cuda")
    bg = torch.tensor([0.1, 0.7, 0.4], device="cuda")
    rs = GaussianRasterizationSettings(**{k: (v.cuda() if torch.is_tensor(v) else v) for k, v in U.settings_kwargs(cam, bg, sh_degree=0).items()})
    r = GaussianRasterizer(rs)
    z = lambda *s: torch.zeros(*s, device="cuda")
    color, radii, invd = r(means3D=z(0, 3), means2D=z(0, 3), opacities=z(0, 1), colors_precomp=z(0, 3), scales=z(0, 3), rotations=z(0, 4))
    assert radii.numel() == 0 and torch.equal(color, bg[:, None, None].expand(3, 32, 48)) and invd.abs().max() == 0
    behind = (cam.camera_center + cam.camera_center / cam.camera_center.norm())[None]
    color, radii, _ = r(means3D=behind, means2D=z(1, 3), opacities=torch.ones(1, 1, device="cuda"),
                        colors_precomp=torch.ones(1, 3, device="cuda"), scales=torch.full((1, 3), 0.1, device="cuda"),
                        rotations=torch.tensor([[1.0, 0, 0, 0]], device="cuda"))
    assert radii[0] == 0 and torch.equal(color, bg[:, None, None].expand(3, 32, 48))
    with pytest.raises(Exception):
        r(means3D=z(1, 3), means2D=z(1, 3), opacities=z(1, 1), scales=z(1, 3), rotations=z(1, 4))
    with pytest.raises(Exception):
        r(means3D=z(1, 3), means2D=z(1, 3), opacities=z(1, 1), colors_precomp=z(1, 3), shs=z(1, 16, 3), scales=z(1, 3), rotations=z(1, 4))
    with pytest.raises(Exception):
        r(means3D=z(1, 3), means2D=z(1, 3), opacities=z(1, 1), colors_precomp=z(1, 3), scales=z(1, 3))
    vis = r.markVisible(torch.cat([behind, z(1, 3)]))
    assert vis.tolist() == [False, True]


def test_long_tile_segments_and_depth_ties():
    """> 4096 splats in one tile (global-memory merge path of the tile sort) and exact depth ties
    (resolved by ascending Gaussian id, like the reference's stable radix sort)."""
    g = torch.Generator().manual_seed(0)
    P = 9000
    means = torch.randn(P, 3, generator=g) * 0.02                 # all inside one or two tiles
    means[: P // 2, 1] = 0.0                                      # thousands of exactly equal depths along the view axis
    sc = syn.random_scene(P, seed=11, scale_lo=0.002, scale_hi=0.01, opacity_lo=0.01, opacity_hi=0.05)
    cam = syn.look_at_camera((0.0, -3.0, 0.0), width=64, height=64, fovx=0.5)
    inputs = dict(means3D=means, opacities=sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    h, o, rep, _ = _check(inputs, U.settings_kwargs(cam, torch.zeros(3)), 64, 64)
    rng = o["details"]["ranges"]
    assert (rng[:, 1] - rng[:, 0]).max() > 4096


def test_capacity_hint_path_equals_sync_path_and_is_deterministic(monkeypatch):
    import diff_gaussian_rasterization as dgr
    sc = syn.random_scene(5000, seed=9, scale_lo=0.01, scale_hi=0.1)
    cam = syn.orbit_camera(2, width=160, height=160, radius=3.0)
    kw = U.settings_kwargs(cam, torch.ones(3))
    monkeypatch.setenv("GMS_SYNC_BINNING", "1")
    a = U.hip_render(_inputs(sc), kw, need_grad=False)
    monkeypatch.setenv("GMS_SYNC_BINNING", "0")
    dgr.clear_capacity_hints()
    b = U.hip_render(_inputs(sc), kw, need_grad=False)            # first call: no hint yet
    c = U.hip_render(_inputs(sc), kw, need_grad=False)            # second call: optimistic path
    assert dgr.last_stats()["capacity_hint"] > 0
    dgr.set_capacity_hint(0, 160, 160, 5000, 10)                  # force an overflow + re-run
    d = U.hip_render(_inputs(sc), kw, need_grad=False)
    for other in (b, c, d):
        assert np.array_equal(a["color"], other["color"]) and np.array_equal(a["radii"], other["radii"])


def test_full_size_mesh_scene_properties():
    """BASELINE-size scene (299 712 mesh-bound Gaussians, 800x800): size-independent properties."""
    from games_hip.model import HipGaussianMeshModel
    from games_hip.render import PipelineParams, render
    model = HipGaussianMeshModel.from_scene(syn.mesh_scene("c2_hotdog_like"), "cuda")
    cam = syn.orbit_camera(0).to("cuda")
    ones = torch.ones_like(model.get_xyz)
    with torch.no_grad():
        white = render(cam, model, PipelineParams(), torch.ones(3, device="cuda"), override_color=ones)["render"]
        black = render(cam, model, PipelineParams(), torch.zeros(3, device="cuda"), override_color=ones)
        # colour 1 everywhere: sum of weights + T_final = 1  =>  white-background image is exactly 1
        assert (white - 1).abs().max().item() <= 1e-4      # float32 sum of up to ~7000 weights per pixel
        cover = black["render"][0]
        assert 0.15 < (cover > 0.5).float().mean().item() < 0.6          # the sphere covers the image centre
        assert (black["radii"] > 0).all()
        # permutation of the splats within faces does not change the image (up to float summation order = none: sorted)
        img1 = render(cam, model, PipelineParams(), torch.ones(3, device="cuda"))["render"]
        img2 = render(cam, model, PipelineParams(), torch.ones(3, device="cuda"))["render"]
        assert torch.equal(img1, img2)                                     # forward is bit-deterministic


def test_full_size_parity_with_oracle_through_the_mesh_op():
    """Whole hot path at BASELINE size: parameters -> K0 -> activations -> rasterizer -> image, and
    image gradient -> parameter gradients, HIP vs (torch-CPU K0 oracle + C rasterizer oracle)."""
    from games_hip.model import HipGaussianMeshModel
    from games_hip.render import PipelineParams, render
    from oracle import gs_oracle, mesh_oracle
    scene = syn.mesh_scene("c2_hotdog_like", state="trained")
    cam_cpu = syn.orbit_camera(3)
    # ---- oracle
    v = scene.vertices.clone().requires_grad_(True)
    a = scene._alpha.clone().requires_grad_(True)
    s = scene._scale.clone().requires_grad_(True)
    op_raw = scene._opacity.clone().requires_grad_(True)
    fdc = scene._features_dc.clone().requires_grad_(True)
    frest = scene._features_rest.clone().requires_grad_(True)
    _, _, xyz, scaling, rot = mesh_oracle.mesh_to_gaussians(v, scene.faces, a, s)
    xyz_a, s_a, r_a, o_a, shs = mesh_oracle.activated(xyz, scaling, rot, op_raw, fdc, frest)
    kw = U.settings_kwargs(cam_cpu, torch.ones(3))
    okw = {k: val for k, val in kw.items() if k not in ("prefiltered", "debug")}
    o = gs_oracle.rasterize(means3D=xyz_a, opacities=o_a, shs=shs, scales=s_a, rotations=r_a, **okw)
    gc = syn.upstream_grad(torch.from_numpy(o.color)) * 1000.0
    g = gs_oracle.backward(o, gc)
    loss = ((xyz_a * torch.from_numpy(g["means3D"])).sum() + (s_a * torch.from_numpy(g["scales"])).sum()
            + (r_a * torch.from_numpy(g["rotations"])).sum() + (o_a * torch.from_numpy(g["opacities"])).sum()
            + (shs * torch.from_numpy(g["sh"])).sum())
    loss.backward()
    # ---- HIP
    model = HipGaussianMeshModel.from_scene(scene, "cuda")

    def run_hip():
        for p_ in (model.vertices, model._alpha, model._scale, model._opacity, model._features_dc, model._features_rest):
            p_.grad = None
        model.update_alpha(); model.prepare_scaling_rot()
        pkg_ = render(cam_cpu.to("cuda"), model, PipelineParams(), torch.ones(3, device="cuda"))
        (pkg_["render"] * gc.cuda()).sum().backward()
        grads = dict(vertices=model.vertices.grad, _alpha=model._alpha.grad, _scale=model._scale.grad, _opacity=model._opacity.grad,
                     f_dc=model._features_dc.grad, f_rest=model._features_rest.grad)
        return dict(pkg=pkg_, grads={k: t.detach().cpu().numpy() for k, t in grads.items()})
    go = dict(vertices=v.grad, _alpha=a.grad, _scale=s.grad, _opacity=op_raw.grad, f_dc=fdc.grad, f_rest=frest.grad)
    def f64_chain():
        """the same chain in double precision (float64 K0 restatement + float64 C rasterizer oracle)"""
        d = lambda t: t.detach().double().clone().requires_grad_(True)
        v6, a6, s6, op6, fdc6, frest6 = d(scene.vertices), d(scene._alpha), d(scene._scale), d(scene._opacity), d(scene._features_dc), d(scene._features_rest)
        _, _, xyz6, scaling6, rot6 = mesh_oracle.mesh_to_gaussians(v6, scene.faces, a6, s6)
        xa, sa, ra, oa, sh6 = mesh_oracle.activated(xyz6, scaling6, rot6, op6, fdc6, frest6)
        o6 = gs_oracle.rasterize(means3D=xa, opacities=oa, shs=sh6, scales=sa, rotations=ra, precision="f64", **okw)
        g6 = gs_oracle.backward(o6, gc.double())
        ((xa * torch.from_numpy(g6["means3D"])).sum() + (sa * torch.from_numpy(g6["scales"])).sum()
         + (ra * torch.from_numpy(g6["rotations"])).sum() + (oa * torch.from_numpy(g6["opacities"])).sum()
         + (sh6 * torch.from_numpy(g6["sh"])).sum()).backward()
        return dict(vertices=v6.grad.numpy(), _alpha=a6.grad.numpy(), _scale=s6.grad.numpy(), _opacity=op6.grad.numpy(),
                    f_dc=fdc6.grad.numpy(), f_rest=frest6.grad.numpy())
    hres, _ = U.assert_grads_both_modes(run_hip, {k: t.numpy() for k, t in go.items()}, f64_chain, where="c2_hotdog_like 800x800 through K0")
    pkg = hres["pkg"]
    h = dict(color=pkg["render"].detach().cpu().numpy(), radii=pkg["radii"].cpu().numpy(), invdepth=pkg["depth"].detach().cpu().numpy())
    ora = dict(color=o.color, radii=o.radii, invdepth=o.invdepth, details=o.state.details())
    # each side ran its own float32 mesh->Gaussian stage: the rasterizer inputs differ by rounding (input_rounding=True)
    rep = U.forward_report(h, ora, 800, 800, input_rounding=True)
    assert rep["radii_unexplained"] == 0 and rep["max_clean"] <= RGB_TOL and rep["amb_frac"] < 0.02, rep
    assert rep["psnr"] > 60.0, rep


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_split_sh_storage_equals_concatenated(deg):
    """SplitSH (DC and REST blocks read / differentiated in place) == the reference's torch.cat path."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, SplitSH
    P = 3000 + deg            # not a multiple of 64: partial last wave
    sc = syn.random_scene(P, seed=20 + deg, scale_lo=0.01, scale_hi=0.1)
    cam = syn.orbit_camera(deg, width=144, height=96, radius=3.0).to("cuda")
    kw = U.settings_kwargs(cam, torch.tensor([0.3, 0.5, 0.7], device="cuda"), sh_degree=deg)
    rs = GaussianRasterizationSettings(**kw)
    base = dict(means3D=sc.means3D.cuda(), opacities=sc.opacities.cuda(), scales=sc.scales.cuda(), rotations=sc.rotations.cuda())
    gen = torch.Generator().manual_seed(1)
    gc = torch.randn(3, 96, 144, generator=gen).cuda()
    outs = []
    for split in (False, True):
        dc = sc.shs[:, :1].contiguous().cuda().requires_grad_(True)
        rest = sc.shs[:, 1:].contiguous().cuda().requires_grad_(True)
        shs = SplitSH(dc, rest) if split else torch.cat((dc, rest), dim=1)
        m2 = torch.zeros(P, 3, device="cuda", requires_grad=True)
        color, radii, _ = GaussianRasterizer(rs)(means2D=m2, shs=shs, **base)
        (color * gc).sum().backward()
        outs.append((color.detach(), dc.grad, rest.grad))
    assert torch.equal(outs[0][0], outs[1][0])
    nb = (deg + 1) ** 2
    torch.testing.assert_close(outs[1][1], outs[0][1], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(outs[1][2], outs[0][2], rtol=1e-4, atol=1e-6)
    if nb < 16:
        assert outs[1][2][:, nb - 1:].abs().max().item() == 0      # coefficients above the active degree: zero gradient
    sp = SplitSH(sc.shs[:, :1], sc.shs[:, 1:])
    assert sp.shape == sc.shs.shape and torch.equal(sp.transpose(1, 2), sc.shs.transpose(1, 2))


def test_debug_and_prefiltered_flags_are_accepted():
    """`debug=True` (pipe.debug, arguments/__init__.py:68) syncs and checks after every kernel; `prefiltered`
    is accepted for API parity.  Results are unchanged."""
    sc = syn.random_scene(1500, seed=31)
    cam = syn.orbit_camera(1, width=96, height=80, radius=3.0)
    kw = U.settings_kwargs(cam, torch.zeros(3))
    a = U.hip_render(_inputs(sc), kw, need_grad=False)
    kw2 = dict(kw, debug=True, prefiltered=True)
    b = U.hip_render(_inputs(sc), kw2, grad_color=np.ones((3, 80, 96), np.float32))
    assert np.array_equal(a["color"], b["color"]) and np.isfinite(b["grads"]["means3D"]).all()


def test_repeated_backward_reuses_the_rezeroed_gradient_records():
    """The [P,16] gradient-record buffer is kept per (device, stream, P) and left zero by the backward kernels
    (grad_accum_rezero): the 1st, 2nd and 3rd backward of the same inputs must all match the oracle, also after a
    backward with a different upstream gradient and after one with the inverse-depth channel in use."""
    sc, cam = syn.random_scene(3000, seed=11, scale_lo=0.01, scale_hi=0.1), syn.orbit_camera(1, width=160, height=112, radius=3.0)
    kw = U.settings_kwargs(cam, torch.tensor([0.1, 0.2, 0.3]))
    W, H = cam.image_width, cam.image_height
    inputs = _inputs(sc)
    o0 = U.oracle_render(inputs, kw)
    gc = syn.upstream_grad(torch.from_numpy(o0["color"])).numpy() * 1000.0
    gdm = np.full((1, H, W), 1e-3, np.float32)
    ref_a = U.oracle_render(inputs, kw, gc, None)
    ref_b = U.oracle_render(inputs, kw, -2.0 * gc, gdm)
    for which in ("a", "a", "b", "a", "b", "a"):
        if which == "a":
            h, ref = U.hip_render(inputs, kw, grad_color=gc, grad_invdepth=None), ref_a
        else:
            h, ref = U.hip_render(inputs, kw, grad_color=-2.0 * gc, grad_invdepth=gdm), ref_b
        gc_w, gd_w = (gc, None) if which == "a" else (-2.0 * gc, gdm)
        U.assert_grads(h["grads"], ref["grads"], lambda: U.oracle_render(inputs, kw, gc_w, gd_w, precision="f64")["grads"],
                       where=f"repeat {which}", excuse=U.excused_rows(ref["details"]),
                       go32acc_fn=lambda: U.f32_realisations(inputs, kw, gc_w, gd_w),
                       alt=U.alt_oracles(inputs, kw, gc_w, gd_w, ref["details"]))


@pytest.mark.parametrize("P", [9000, 30000])
def test_deep_tiles_take_the_merge_path_sort_from_the_second_frame(P):
    """Tiles deeper than 8192 keys: frame 1 sorts them with the one-block fallback (no history), later frames with
    the multi-block merge-path passes sized from the previous frame's deepest tile.  Every frame must match the oracle
    (depth ties by id included), and a shallow frame in between must not disturb anything."""
    g = torch.Generator().manual_seed(2)
    means = torch.randn(P, 3, generator=g) * 0.015
    means[: P // 3, 1] = 0.0                                      # exact depth ties
    sc = syn.random_scene(P, seed=13, scale_lo=0.002, scale_hi=0.008, opacity_lo=0.005, opacity_hi=0.03)
    cam = syn.look_at_camera((0.0, -3.0, 0.0), width=48, height=48, fovx=0.5)
    kw = U.settings_kwargs(cam, torch.zeros(3))
    inputs = dict(means3D=means, opacities=sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    o = U.oracle_render(inputs, kw)
    rng = o["details"]["ranges"]
    assert (rng[:, 1] - rng[:, 0]).max() > 8192
    shallow = syn.random_scene(500, seed=3)
    sh_in, sh_kw = _inputs(shallow), U.settings_kwargs(syn.orbit_camera(0, width=48, height=48), torch.zeros(3))
    for frame in range(4):
        if frame == 2:
            U.hip_render(sh_in, sh_kw, need_grad=False)           # a shallow frame resets the pass count
        h = U.hip_render(inputs, kw, need_grad=False)
        rep = U.forward_report(h, o, 48, 48)
        assert rep["radii_unexplained"] == 0 and rep["max_clean"] <= RGB_TOL and rep["max_amb"] <= 0.02, (frame, rep)


@pytest.mark.parametrize("env", [
    {"GMS_INLINE_SCAN": "0"},
    # micro-tile compositing (the default, blend_micro.hip): segment lengths, forced two-phase products, entries per trip
    {"GMS_SEG_LEN": "64"}, {"GMS_SEG_LEN": "128", "GMS_DEEP": "1"}, {"GMS_SEG_LEN": "192"}, {"GMS_TRIP": "4", "GMS_TRIP_BWD": "1"},
    {"GMS_TRIP": "1", "GMS_TRIP_BWD": "4"}, {"GMS_UNIT_RUN": "1"}, {"GMS_SYNC_BINNING": "1"}, {"GMS_BINDING": "ctypes"},
    {"GMS_BINDING": "ctypes", "GMS_SYNC_BINNING": "1"},
    # the quadrant-wave kernels of blend.hip (GMS_MICRO=0) with their own knobs
    {"GMS_MICRO": "0"}, {"GMS_MICRO": "0", "GMS_SEG_LEN": "64"}, {"GMS_MICRO": "0", "GMS_SEG_LEN": "256"},
    {"GMS_MICRO": "0", "GMS_SEG_LEN": "512", "GMS_DEEP": "1"}, {"GMS_MICRO": "0", "GMS_TRIP": "2", "GMS_TRIP_BWD": "2"},
    {"GMS_MICRO": "0", "GMS_FWD_WPB": "4", "GMS_BWD_WPB": "4"},
    {"GMS_MICRO": "0", "GMS_FWD_WPB": "4", "GMS_BWD_WPB": "4", "GMS_SEG_LEN": "256", "GMS_DEEP": "1"}])
def test_tuning_knobs_do_not_change_results(env):
    """The knobs of INTEGRATION.md section 6 are read once per process: run the parity check in a child process per
    setting (both compositing implementations; segment lengths other than the default, forced two-phase products, entries
    per trip, the four-waves-per-block layout of the quadrant kernels, no XCD run interleave, synchronous binning)."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, os, torch\n"
        "sys.path.insert(0, os.path.join(%r, 'tests'))\n"
        "import conftest\n"
        "import test_gpu_raster as T\n"
        "from games_hip import synthetic as syn\n"
        "import _util as U\n"
        "sc, cam = syn.random_scene(6000, seed=21, scale_lo=0.01, scale_hi=0.15), syn.orbit_camera(2, width=144, height=112, radius=2.5)\n"
        "T._check(T._inputs(sc), U.settings_kwargs(cam, torch.tensor([0.3, 0.1, 0.2])), 144, 112)\n"
        "print('knob ok')\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "knob ok" in r.stdout, (env, r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.parametrize("env", [{}, {"GMS_MICRO": "1"}, {"GMS_MICRO": "0"}, {"GMS_MICRO": "1", "GMS_SEG_LEN": "128"}])
def test_very_deep_frames_keep_a_segment_length_their_kernels_support(env):
    """6 000 large splats on 9 tiles: ~6 000 list entries per tile, beyond SEG_VERY_DEEP_PER_TILE.  The second call has the
    capacity hint that moves very deep frames to 512-entry segments -- which only the quadrant kernels can take (the
    micro-tile kernels index a unit's entries with one byte): auto mode must switch kernels with the segment length,
    GMS_MICRO=1 must keep its own L.  (A forced-micro run at config-5 size failed on exactly this in round 3.)"""
    import os
    import subprocess
    import sys
    code = (
        "import sys, os, torch\n"
        "sys.path.insert(0, os.path.join(%r, 'tests'))\n"
        "import conftest\n"
        "import test_gpu_raster as T\n"
        "from games_hip import synthetic as syn\n"
        "import _util as U\n"
        "import diff_gaussian_rasterization as dgr\n"
        "sc, cam = syn.random_scene(6000, seed=33, extent=0.5, scale_lo=0.08, scale_hi=0.4, opacity_lo=0.02, opacity_hi=0.3), syn.orbit_camera(3, width=48, height=48, radius=2.0)\n"
        "for call in range(3):\n"
        "    h, o, rep, g = T._check(T._inputs(sc), U.settings_kwargs(cam, torch.tensor([0.3, 0.1, 0.2])), 48, 48)\n"
        "assert o['N'] > 2048 * 9 * 1.3, o['N']\n"
        "assert dgr.last_stats()['capacity_hint'] > 2048 * 9\n"
        "print('deep ok')\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=900,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "deep ok" in r.stdout, (env, r.stdout[-2000:], r.stderr[-3000:])


def huge_thin_scene(P=120, seed=7, S=16.0):
    """Splats ~1 000 px long (sigma) and a fraction of a pixel wide whose TIPS lie inside a 192x192 image, centres up to
    3 500 px outside it: beyond the tip the float32 exponent is a difference of terms ~10^7 and accepts pixels outside the
    exact ellipse's bounding box (tests/test_filter_emulation.py; on this scene the exact box drops 16 (splat, 4x4 block)
    pairs some pixel accepts)."""
    rng = np.random.default_rng(seed)
    phi = rng.uniform(0, np.pi, P)
    axis = np.stack([np.cos(phi), np.zeros(P), np.sin(phi)], 1)
    t = rng.uniform(2.7, 3.5, P) * S
    off = rng.normal(0, 0.4, (P, 3)); off[:, 1] = rng.uniform(-0.5, 0.5, P)
    means = -(t[:, None] * axis) + off
    q = np.stack([np.cos(-phi / 2), np.zeros(P), np.sin(-phi / 2), np.zeros(P)], 1)
    scales = np.stack([np.full(P, S), np.full(P, 0.002), rng.uniform(0.002, 0.02, P)], 1)
    g = torch.Generator().manual_seed(seed)
    return dict(means3D=torch.tensor(means, dtype=torch.float32), opacities=torch.tensor(rng.uniform(0.3, 0.9, (P, 1)), dtype=torch.float32),
                shs=torch.randn(P, 16, 3, generator=g) * 0.3, scales=torch.tensor(scales, dtype=torch.float32),
                rotations=torch.tensor(q, dtype=torch.float32))


@pytest.mark.parametrize("env", [{}, {"GMS_MICRO": "0"}])
def test_huge_thin_splats_keep_the_pixels_their_float32_exponent_accepts(env):
    """Both compositing implementations against the oracle on `huge_thin_scene` (forward image, radii, inverse depth; the
    gradients of such splats are float32 noise on both sides and are only required to be finite)."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, os, torch, numpy as np\n"
        "sys.path.insert(0, os.path.join(%r, 'tests'))\n"
        "import conftest\n"
        "import test_gpu_raster as T\n"
        "from games_hip import synthetic as syn\n"
        "import _util as U\n"
        "inputs = T.huge_thin_scene()\n"
        "cam = syn.look_at_camera((0.0, -3.0, 0.0), width=192, height=192, fovx=0.9)\n"
        "kw = U.settings_kwargs(cam, torch.tensor([0.1, 0.2, 0.3]), sh_degree=1)\n"
        "o = U.oracle_render(inputs, kw)\n"
        "gc = syn.upstream_grad(torch.from_numpy(o['color'])).numpy() * 1000.0\n"
        "for call in range(2):\n"
        "    h = U.hip_render(inputs, kw, grad_color=gc)\n"
        "    rep = U.forward_report(h, o, 192, 192)\n"
        "    assert rep['radii_unexplained'] == 0 and rep['max_clean'] <= 1e-4 and rep['max_invdepth_clean'] <= 1e-4 and rep['max_amb'] <= 0.02, rep\n"
        "    assert all(np.isfinite(v).all() for v in h['grads'].values() if v is not None)\n"
        "assert int((o['radii'] > 3000).sum()) > 20 and o['N'] > 5000\n"
        "print('huge ok', rep)\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "huge ok" in r.stdout, (env, r.stdout[-2000:], r.stderr[-3000:])


def moderate_thin_scene(P=3000, seed=11, W=512, fovx=0.9, dist=3.0):
    """Minimum-width splats (a third of a pixel before the 0.3 dilation) 45 ... 110 px long (sigma), opacities 0.006 ... 1, ONE TIP
    INSIDE the image: bounding-box half extents mostly 100 ... 260 px -- the regime in which round 3's exact bounding box still lost
    fringe pixels to the float32 noise of the per-pixel exponent (tests/test_filter_emulation.py, tests/test_gpu_cull.py)."""
    rng = np.random.default_rng(seed)
    ppu = W / (2 * np.tan(fovx / 2)) / dist          # pixels per world unit in the plane through the origin
    s1_px = np.exp(rng.uniform(np.log(45.0), np.log(110.0), P))
    op = np.exp(rng.uniform(np.log(0.006), np.log(1.0), P))
    phi = rng.uniform(0, np.pi, P)
    axis = np.stack([np.cos(phi), np.zeros(P), np.sin(phi)], 1)
    t_px = np.sqrt(np.maximum(2 * np.log(255 * op), 0.05)) * s1_px * rng.uniform(0.97, 1.05, P)      # centre -> tip (pixels)
    tip = np.stack([rng.uniform(-0.45, 0.45, P) * W / ppu, np.zeros(P), rng.uniform(-0.45, 0.45, P) * W / ppu], 1)
    sign = np.where(rng.uniform(size=P) < 0.5, -1.0, 1.0)
    means = tip - sign[:, None] * (t_px / ppu)[:, None] * axis
    means[:, 1] = rng.uniform(-0.3, 0.3, P)
    q = np.stack([np.cos(-phi / 2), np.zeros(P), np.sin(-phi / 2), np.zeros(P)], 1)
    scales = np.stack([s1_px / ppu, np.full(P, 0.0004), np.full(P, 0.0004)], 1)
    g = torch.Generator().manual_seed(seed)
    return dict(means3D=torch.tensor(means, dtype=torch.float32), opacities=torch.tensor(op[:, None], dtype=torch.float32),
                shs=torch.randn(P, 16, 3, generator=g) * 0.3, scales=torch.tensor(scales, dtype=torch.float32),
                rotations=torch.tensor(q, dtype=torch.float32))


@pytest.mark.parametrize("env", [{}, {"GMS_MICRO": "0"}])
def test_moderate_thin_splats_with_extents_of_144_to_253_px(env):
    """Both compositing implementations against the oracle on `moderate_thin_scene`: image, radii, inverse depth within the
    suite's tolerances on every unflagged pixel (the gradients of such splats are float32 noise on both sides: finite)."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, os, torch, numpy as np\n"
        "sys.path.insert(0, os.path.join(%r, 'tests'))\n"
        "import conftest\n"
        "import test_gpu_raster as T\n"
        "from games_hip import synthetic as syn\n"
        "import _util as U\n"
        "inputs = T.moderate_thin_scene()\n"
        "cam = syn.look_at_camera((0.0, -3.0, 0.0), width=512, height=512, fovx=0.9)\n"
        "kw = U.settings_kwargs(cam, torch.tensor([0.1, 0.2, 0.3]), sh_degree=1)\n"
        "o = U.oracle_render(inputs, kw)\n"
        "co = o['details']['conic_op']; det = co[:, 0] * co[:, 2] - co[:, 1] ** 2\n"
        "thr = np.maximum(2 * np.log(255 * co[:, 3]), 0)\n"
        "ext = np.maximum(np.sqrt(co[:, 2] / det * thr), np.sqrt(co[:, 0] / det * thr))\n"
        "assert int(((ext >= 144) & (ext <= 253)).sum()) > 800 and o['N'] > 500000, (ext, o['N'])\n"
        "gc = syn.upstream_grad(torch.from_numpy(o['color'])).numpy() * 1000.0\n"
        "for call in range(2):\n"
        "    h = U.hip_render(inputs, kw, grad_color=gc)\n"
        "    rep = U.forward_report(h, o, 512, 512)\n"
        "    assert rep['radii_unexplained'] == 0 and rep['max_clean'] <= 1e-4 and rep['max_invdepth_clean'] <= 1e-4 and rep['max_amb'] <= 0.02, rep\n"
        "    assert rep['amb_frac'] < 0.01, rep\n"
        "    assert all(np.isfinite(v).all() for v in h['grads'].values() if v is not None)\n"
        "print('moderate ok', rep)\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "moderate ok" in r.stdout, (env, r.stdout[-2000:], r.stderr[-3000:])


def test_visibility_filter_from_the_preprocess_kernel_equals_radii_positive():
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    sc, cam = syn.random_scene(5000, seed=31, scale_lo=0.01, scale_hi=0.1), syn.orbit_camera(3, width=128, height=96, radius=1.2)
    kw = U.settings_kwargs(cam, torch.zeros(3))
    for k in ("bg", "viewmatrix", "projmatrix", "campos"):
        kw[k] = kw[k].cuda().float()
    r = GaussianRasterizer(GaussianRasterizationSettings(**kw))
    t = {k: v.cuda().float() for k, v in _inputs(sc).items()}
    _, radii, _ = r(means3D=t["means3D"], means2D=torch.zeros_like(t["means3D"]), opacities=t["opacities"], shs=t["shs"],
                    scales=t["scales"], rotations=t["rotations"])
    assert r.visibility_filter.dtype == torch.bool and torch.equal(r.visibility_filter, radii > 0)
    assert 0 < int(r.visibility_filter.sum()) < 5000          # the close camera culls some Gaussians behind the near plane


def test_unit_count_overflow_reruns_with_full_size_launches():
    """The forward sizes its blend launches from a decaying maximum of the unit counts of recent frames OF THE SAME SHAPE
    (device, stream, W, H, P).  After many frames from a distant camera (every splat inside one tile: one unit per tile) that
    estimate has decayed, so a close-up frame has more units than blocks were launched for although the binning buffer is
    large enough: it must be re-run transparently and still match the oracle (image and gradients)."""
    import diff_gaussian_rasterization as dgr
    sc = syn.random_scene(20000, seed=41, scale_lo=0.01, scale_hi=0.15)
    far = syn.orbit_camera(1, width=416, height=400, radius=60.0)             # 650 tiles
    near = syn.orbit_camera(1, width=416, height=400, radius=2.5)
    inputs = _inputs(sc)
    kw_far, kw = U.settings_kwargs(far, torch.tensor([0.2, 0.3, 0.1])), U.settings_kwargs(near, torch.tensor([0.2, 0.3, 0.1]))
    o = U.oracle_render(inputs, kw)
    gc = syn.upstream_grad(torch.from_numpy(o["color"])).numpy() * 1000.0
    o = U.oracle_render(inputs, kw, gc, None)
    U.hip_render(inputs, kw, need_grad=False)                         # first call of this shape: exact sizes
    units_near, n_near = dgr.last_stats()["num_units"], dgr.last_stats()["num_rendered"]
    for _ in range(200):                                              # 0.97^200 < 1 %: the unit estimate decays to the far frames'
        U.hip_render(inputs, kw_far, need_grad=False)
    units_far = dgr.last_stats()["num_units"]
    assert units_near > 1.25 * units_far + 64 + 512, (units_near, units_far)      # more units than the padded optimistic launch
    dgr.set_capacity_hint(0, 416, 400, 20000, n_near)                 # the binning buffer itself is large enough
    h = U.hip_render(inputs, kw, grad_color=gc, grad_invdepth=None)
    rep = U.forward_report(h, o, 416, 400)
    assert rep["radii_unexplained"] == 0 and rep["max_clean"] <= RGB_TOL and rep["max_amb"] <= 0.02, rep
    U.assert_grads(h["grads"], o["grads"], lambda: U.oracle_render(inputs, kw, gc, None, precision="f64")["grads"], where="unit overflow",
                   excuse=U.excused_rows(o["details"]), go32acc_fn=lambda: U.f32_realisations(inputs, kw, gc, None),
                   alt=U.alt_oracles(inputs, kw, gc, None, o["details"]))


def test_config5_size_parity_deep_tiles_two_phase_products():
    """BASELINE config 5 size: 997 600 mesh-bound Gaussians (FLAME-like F = 9 976 x S = 100), 1024x1024, ~12 M instances,
    tiles up to ~26 k keys deep.  Exercises at full size what the toy scenes cannot: >= 2 multi-block merge-path sort
    passes, the two-phase transmittance products with the per-tile dead check (taken only when capacity > 2 L T) and
    dozens of segments per tile.  Stage by stage on IDENTICAL inputs:
      (1) K0 at this size against the torch restatement;
      (2) the rasterizer (image, radii, n_contrib-driven backward: gradients of all five inputs) against the OpenMP
          oracle, both fed the RESTATEMENT's Gaussians -- frame 0 without history (synchronous sizes, one-block sort
          fallback), frames 1-2 on the capacity / unit / sort-pass hints;
      (3) K0 backward at this size with the oracle's rasterizer gradients as upstream."""
    from diff_gaussian_rasterization import last_stats
    from games_hip.mesh_op import mesh_to_gaussians
    from oracle import mesh_oracle
    scene = syn.mesh_scene("c5_flame_like_1m", state="trained")
    size = scene.meta["image"]
    cam = syn.orbit_camera(2, width=size, height=size)
    leaf = lambda t: t.clone().requires_grad_(True)
    v, a, s = leaf(scene.vertices), leaf(scene._alpha), leaf(scene._scale)
    _, _, xyz, scaling, rot = mesh_oracle.mesh_to_gaussians(v, scene.faces, a, s)
    xyz_a, s_a, r_a, o_a, shs = mesh_oracle.activated(xyz, scaling, rot, scene._opacity, scene._features_dc, scene._features_rest)
    # (1) K0 forward
    vg, ag, sg = scene.vertices.cuda().requires_grad_(True), scene._alpha.cuda().requires_grad_(True), scene._scale.cuda().requires_grad_(True)
    _, xyz_h, scaling_h, rot_h, sact_h, runit_h = mesh_to_gaussians(vg, scene.faces.cuda(), ag, sg, "relu", fused_activations=True)
    for got, ref, tol in ((xyz_h, xyz, 1e-5), (sact_h, s_a, 1e-5), (runit_h, r_a, 1e-4)):
        d = (got.detach().cpu() - ref.detach()).abs()
        assert float(d.max()) <= tol * float(ref.detach().abs().max()) + 1e-12
    # (2) rasterizer on identical inputs
    inputs = dict(means3D=xyz_a.detach(), opacities=o_a.detach(), shs=shs.detach(), scales=s_a.detach(), rotations=r_a.detach())
    kw = U.settings_kwargs(cam, torch.ones(3))
    o = U.oracle_render(inputs, kw)
    gc = syn.upstream_grad(torch.from_numpy(o["color"])).numpy() * 1000.0
    o = U.oracle_render(inputs, kw, gc, None)
    det = o["details"]
    depth = int((det["ranges"][:, 1] - det["ranges"][:, 0]).max())
    assert det["N"] > 2 * 256 * (size // 16) ** 2 and depth > 2 * 8192, (det["N"], depth)       # deep path + >= 2 merge-path passes
    lazy = dict(go64=U._memo(lambda: U.oracle_render(inputs, kw, gc, None, precision="f64")["grads"]),
                go32=U._memo(lambda: U.f32_realisations(inputs, kw, gc, None)))
    alt = U.alt_oracles(inputs, kw, gc, None, det)
    alt = (alt[0], U._memo(alt[1]))
    # PRIMARY gate: deterministic-reduction mode (quadrant kernels, per-(instance, quadrant) partial records), strict criterion
    import diff_gaussian_rasterization as dgr
    dgr.set_deterministic(True)
    try:
        h = U.hip_render(inputs, kw, grad_color=gc, grad_invdepth=None)
    finally:
        dgr.set_deterministic(False)
    U.assert_grads(h["grads"], o["grads"], lazy["go64"], where=f"c5_flame_like_1m {size}x{size} [deterministic, strict]", excuse=U.excused_rows(det),
                   go32acc_fn=lazy["go32"], alt=alt, strict=True)
    # SECONDARY: the float-atomics mode, three frames (no history / hints / hints)
    for frame in range(3):
        h = U.hip_render(inputs, kw, grad_color=gc, grad_invdepth=None)
        rep = U.forward_report(h, o, size, size)
        assert rep["radii_unexplained"] == 0 and rep["max_clean"] <= RGB_TOL and rep["amb_frac"] < 0.01 and rep["max_amb"] <= 0.02, (frame, rep)
        assert last_stats()["num_rendered"] == det["N"]
        U.assert_grads(h["grads"], o["grads"], lazy["go64"],
                       where=f"c5_flame_like_1m {size}x{size} frame {frame} [atomics]", excuse=U.excused_rows(det),
                       go32acc_fn=lazy["go32"], alt=alt)
        assert float(U.excused_rows(det).mean()) < 0.02
    # (3) K0 backward at this size, upstream = the oracle's rasterizer gradients
    og = o["grads"]
    gx, gs, gr = (torch.from_numpy(og[k]) for k in ("means3D", "scales", "rotations"))
    ((xyz_a * gx).sum() + (s_a * gs).sum() + (r_a * gr).sum()).backward()
    ((xyz_h * gx.cuda()).sum() + (sact_h * gs.cuda()).sum() + (runit_h * gr.cuda()).sum()).backward()

    def f64_k0():
        d = lambda t: t.detach().double().clone().requires_grad_(True)
        v6, a6, s6 = d(scene.vertices), d(scene._alpha), d(scene._scale)
        _, _, xyz6, scaling6, rot6 = mesh_oracle.mesh_to_gaussians(v6, scene.faces, a6, s6)
        ((xyz6 * gx.double()).sum() + (torch.exp(scaling6) * gs.double()).sum()
         + (torch.nn.functional.normalize(rot6) * gr.double()).sum()).backward()
        return dict(vertices=v6.grad.numpy(), _alpha=a6.grad.numpy(), _scale=s6.grad.numpy())
    U.assert_grads(dict(vertices=vg.grad.cpu().numpy(), _alpha=ag.grad.cpu().numpy(), _scale=sg.grad.cpu().numpy()),
                   dict(vertices=v.grad.numpy(), _alpha=a.grad.numpy(), _scale=s.grad.numpy()), f64_k0, where="K0 backward at c5 size")


def test_two_streams_interleaving_two_scene_sizes_match_the_serial_run():
    """Library state that outlives a call (tile counters, capacity / unit / sort-pass hints) is keyed by (device, stream,
    W, H, P): two torch streams in one process, each rendering its own scene (different P and image size) in alternation and
    without synchronising in between, must give bit-identical images and radii to the same scenes rendered serially."""
    cases = [(syn.random_scene(7000, seed=51, scale_lo=0.01, scale_hi=0.12), syn.orbit_camera(1, width=272, height=208, radius=2.8)),
             (syn.random_scene(1800, seed=52, scale_lo=0.02, scale_hi=0.2), syn.orbit_camera(4, width=96, height=144, radius=3.2))]
    kws = [U.settings_kwargs(cam, torch.tensor([0.1, 0.2, 0.3])) for _, cam in cases]
    serial = [U.hip_render(_inputs(sc), kw, need_grad=False) for (sc, _), kw in zip(cases, kws)]
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    dev = torch.device("cuda")
    tens, rss = [], []
    for (sc, _), kw in zip(cases, kws):
        tens.append({k: v.to(dev).float() for k, v in _inputs(sc).items()})
        kwd = dict(kw)
        for k in ("bg", "viewmatrix", "projmatrix", "campos"):
            kwd[k] = kwd[k].to(dev).float()
        rss.append(GaussianRasterizationSettings(**kwd))
    torch.cuda.synchronize()
    outs = [[], []]
    for rep in range(6):                                # frames 0: synchronous sizes; 1..: optimistic hints, all in flight
        for k in (0, 1):
            with torch.cuda.stream(streams[k]):
                t = tens[k]
                color, radii, _ = GaussianRasterizer(rss[k])(means3D=t["means3D"], means2D=torch.zeros_like(t["means3D"]),
                                                             opacities=t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
                outs[k].append((color, radii))
    torch.cuda.synchronize()
    for k in (0, 1):
        for color, radii in outs[k]:
            assert np.array_equal(color.cpu().numpy(), serial[k]["color"]) and np.array_equal(radii.cpu().numpy(), serial[k]["radii"])


def test_debug_flag_dumps_the_inputs_of_a_failing_forward(tmp_path, monkeypatch):
    """Upstream behaviour with pipe.debug: a failing forward leaves `snapshot_fw.dump` (CPU copies of its arguments)."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    monkeypatch.chdir(tmp_path)
    sc, cam = syn.random_scene(300, seed=1), syn.orbit_camera(0, width=48, height=48)
    kw = U.settings_kwargs(cam, torch.zeros(3), sh_degree=5)           # degree 5 needs 36 coefficients: rejected (16 stored)
    kw["debug"] = True
    for k in ("bg", "viewmatrix", "projmatrix", "campos"):
        kw[k] = kw[k].cuda().float()
    t = {k: v.cuda().float() for k, v in _inputs(sc).items()}
    with pytest.raises(RuntimeError):
        GaussianRasterizer(GaussianRasterizationSettings(**kw))(means3D=t["means3D"], means2D=torch.zeros_like(t["means3D"]),
                                                                opacities=t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    snap = torch.load(str(tmp_path / "snapshot_fw.dump"), weights_only=False)
    assert torch.equal(snap[0], sc.means3D) and snap[8][8] == 5
