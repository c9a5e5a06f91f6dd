"""GPU: TRAINING frames rendered straight from the mesh (GmsRasterForwardArgs.mesh + mesh_out_*, ABI 6; games_hip.model.HipMeshMixin.hip_defer_k0).
train.py:154-157 calls update_alpha() / prepare_scaling_rot() after every optimizer step and train.py:100 renders next: with the K0 deferred,
the face -> Gaussian arithmetic runs inside the rasterizer's preprocess thread and ONE autograd node carries the gradients from the image
to (vertices, _alpha, _scale, _opacity, SH).  The frame, every gradient and the model's derived attributes must equal the two-node route
(K0 launch, then the rasterizer): bit for bit in the forward and -- in the deterministic-reduction mode -- in the backward."""
import pytest
import torch

from games_hip import synthetic as syn

pytestmark = pytest.mark.gpu
PARAMS = ("vertices", "_alpha", "_scale", "_opacity", "_features_dc", "_features_rest")


def _model(name="small", splats=None):
    from games_hip.model import HipGaussianMeshModel
    return HipGaussianMeshModel.from_scene(syn.mesh_scene(name, splats=splats), "cuda")


def _step(model, cam, bg, defer):
    from games_hip.render import PipelineParams, render
    model.hip_defer_k0 = defer
    for n in PARAMS:
        getattr(model, n).grad = None
    model.update_alpha(); model.prepare_scaling_rot()
    out = render(cam, model, PipelineParams(), bg)
    img = out["render"]
    # renderer/gaussian_renderer/__init__.py:108: on either route the filter comes out of the preprocess kernel, not out of a comparison launch
    assert out["visibility_filter"].dtype == torch.bool and torch.equal(out["visibility_filter"], out["radii"] > 0)
    (img * ((img.detach() - 0.5) / img.numel() * 1000.0)).sum().backward()
    g = {n: getattr(model, n).grad.detach().clone() for n in PARAMS}
    g["viewspace"] = out["viewspace_points"].grad.detach().clone()
    return img.detach().clone(), out["radii"].clone(), g


@pytest.mark.parametrize("det", [True, False])
@pytest.mark.parametrize("size", [128, 203])
def test_fused_training_frame_equals_the_two_node_graph(det, size):
    import diff_gaussian_rasterization as dgr
    model = _model()
    cam = syn.orbit_camera(2, width=size, height=size - 5).to("cuda")
    bg = torch.tensor([0.9, 0.7, 0.3], device="cuda")
    was = dgr.deterministic()
    dgr.set_deterministic(det)
    try:
        _step(model, cam, bg, False)                                    # (the first K0 of a model's life is always eager)
        img0, radii0, g0 = _step(model, cam, bg, False)
        before = dgr._C.last_stats()["num_rendered"]
        img1, radii1, g1 = _step(model, cam, bg, True)
        assert model.__dict__.get("_hip_pending") is True               # K0 really was deferred: no eager launch happened
        assert dgr._C.last_stats()["num_rendered"] == before
    finally:
        dgr.set_deterministic(was)
        model.hip_defer_k0 = False
    assert torch.equal(img1, img0) and torch.equal(radii1, radii0)       # the forward has no atomics: bit for bit in both modes
    for k in g0:
        if det:
            assert torch.equal(g1[k], g0[k]), k                          # fixed summation order: bit for bit
        else:
            scale = float(g0[k].abs().max())
            assert float((g1[k] - g0[k]).abs().max()) <= 2e-5 * scale + 1e-12, k      # two runs differ by the order of the float atomics
        assert float(g0[k].abs().max()) > 0, k


def test_deferred_values_are_materialised_for_every_reader_and_served_from_the_frame_under_no_grad():
    from games_hip.render import PipelineParams, render
    model = _model()
    cam = syn.orbit_camera(1, width=96, height=96).to("cuda")
    bg = torch.ones(3, device="cuda")
    model.update_alpha(); model.prepare_scaling_rot()
    model.hip_defer_k0 = True
    with torch.no_grad():
        model._alpha.add_(0.01 * torch.randn_like(model._alpha))        # "an optimizer step"
    model.update_alpha(); model.prepare_scaling_rot()                    # deferred
    assert model.__dict__.get("_hip_pending") is True
    stale = model._xyz.detach().clone()
    out = render(cam, model, PipelineParams(), bg)                       # the frame derives the Gaussians itself
    assert out["render"].requires_grad and out["viewspace_points"].requires_grad
    with torch.no_grad():
        xyz_frame = model.get_xyz                                        # under no_grad: what the frame derived, no K0 launch
        assert model.__dict__.get("_hip_pending") is True and not torch.equal(xyz_frame, stale)
        sc_frame, rot_frame, op_frame = model.get_scaling, model.get_rotation, model.get_opacity
    xyz = model.get_xyz                                                  # grad mode: the K0 launch (a differentiable tensor)
    assert model.__dict__.get("_hip_pending") is None and xyz.requires_grad
    assert torch.equal(xyz.detach(), xyz_frame) and torch.equal(model._xyz.detach(), xyz_frame)
    assert torch.equal(model.get_scaling.detach(), sc_frame) and torch.equal(model.get_rotation.detach(), rot_frame)
    assert torch.equal(model.get_opacity.detach(), op_frame)
    # the python-stage flags take the two-node route (the SH ramp of training does not: test_fused_training_frame_during_the_sh_ramp)
    model.update_alpha(); model.prepare_scaling_rot()
    pipe = PipelineParams(); pipe.convert_SHs_python = True
    out2 = render(cam, model, pipe, bg)
    assert model.__dict__.get("_hip_pending") is None and out2["render"].requires_grad
    model.hip_defer_k0 = False


@pytest.mark.parametrize("degree", [0, 1, 2])
def test_fused_training_frame_during_the_sh_ramp(degree):
    """train.py:86-87 raises the active SH degree every 1 000 iterations: below 3 the frame is still rendered straight from the mesh
    (`preprocess_fwd_dma_kernel<true, degree>`: the stored rows come in whole, the active bands are evaluated) and the backward leaves
    the inactive bands' gradients zero, exactly as the two-node graph does -- bit for bit in deterministic mode."""
    import diff_gaussian_rasterization as dgr
    model = _model()
    cam = syn.orbit_camera(1, width=160, height=128).to("cuda")
    bg = torch.tensor([0.1, 0.5, 0.8], device="cuda")
    was = dgr.deterministic()
    dgr.set_deterministic(True)
    try:
        model.active_sh_degree = degree
        _step(model, cam, bg, False)
        img0, radii0, g0 = _step(model, cam, bg, False)
        img1, radii1, g1 = _step(model, cam, bg, True)
        assert model.__dict__.get("_hip_pending") is True               # the frame took the fused route
    finally:
        dgr.set_deterministic(was)
        model.active_sh_degree = 3
        model.hip_defer_k0 = False
    assert torch.equal(img1, img0) and torch.equal(radii1, radii0)
    for k in g0:
        assert torch.equal(g1[k], g0[k]), k
    nb = (degree + 1) ** 2 - 1
    assert float(g1["_features_rest"][:, nb:].abs().max()) == 0.0          # bands above the active degree: untouched
    if nb:
        assert float(g1["_features_rest"][:, :nb].abs().max()) > 0.0


def test_training_loop_walks_the_same_trajectory_with_and_without_the_deferred_k0(monkeypatch):
    """games_hip/train.py (= train.py:39-157) for 30 iterations, deterministic-reduction mode: identical parameters at the end."""
    import diff_gaussian_rasterization as dgr
    from games_hip.render import PipelineParams
    from games_hip.train import OptimizationParamsMesh, training
    import random

    def run(fused):
        monkeypatch.setenv("GMS_TRAIN_FUSED", "1" if fused else "0")
        random.seed(0); torch.manual_seed(0)
        model = _model()                                                  # (trained-like state, SH degree 3: the fused route is taken)
        opt = OptimizationParamsMesh(iterations=30, vertices_lr=0.00016)
        model.training_setup(vertices_lr=opt.vertices_lr, alpha_lr=opt.alpha_lr, feature_lr=opt.feature_lr, opacity_lr=opt.opacity_lr,
                             scaling_lr=opt.scaling_lr, fused=True)
        cams = []
        for k in range(4):
            c = syn.orbit_camera(k, width=96, height=96).to("cuda")
            c.original_image = torch.rand(3, 96, 96, device="cuda", generator=torch.Generator(device="cuda").manual_seed(k))
            cams.append(c)
        model.update_alpha(); model.prepare_scaling_rot()
        training(model, cams, opt, PipelineParams(), torch.ones(3, device="cuda"))
        return {n: getattr(model, n).detach().clone() for n in PARAMS}, model

    was = dgr.deterministic()
    dgr.set_deterministic(True)
    try:
        a, _ = run(False)
        b, mb = run(True)
    finally:
        dgr.set_deterministic(was)
    assert getattr(mb, "hip_defer_k0", False) is True
    for n in PARAMS:
        assert torch.equal(a[n], b[n]), n


@pytest.mark.parametrize("splats", [1, 2, 3, 4])
@pytest.mark.parametrize("size", [128, 203])
def test_mesh_backward_inside_preprocess_bwd_equals_the_mesh_backward_launch(size, splats):
    """ABI 8 (GmsRasterBackwardArgs.mesh): the thread of a Gaussian carries dL/dxyz / dL/dscale / dL/drot / dL/dopacity on through the
    face -> Gaussian parameterization from registers.  Same gradients as preprocess_bwd + mesh_bwd_fused up to the order of the float
    atomics on the vertices (the splats of a face are summed inside their wave first; a face of three splats that lies inside one wave
    adds x, y, z of a corner from its three lanes, any other run from its first lane: 1 .. 4 splats per face put the faces at every
    offset against the waves); deterministic mode keeps the two launches."""
    import diff_gaussian_rasterization as dgr
    model = _model(splats=splats)
    cam = syn.orbit_camera(3, width=size, height=size - 7).to("cuda")
    bg = torch.tensor([0.2, 0.9, 0.4], device="cuda")
    _step(model, cam, bg, False)
    was = dgr._C.fused_mesh_backward()
    try:
        dgr._C.set_fused_mesh_backward(False)
        img0, _, g0 = _step(model, cam, bg, True)
        dgr._C.set_fused_mesh_backward(True)
        img1, _, g1 = _step(model, cam, bg, True)
    finally:
        dgr._C.set_fused_mesh_backward(was)
        model.hip_defer_k0 = False
    assert torch.equal(img1, img0)
    for k in g0:
        scale = float(g0[k].abs().max())
        assert scale > 0 and float((g1[k] - g0[k]).abs().max()) <= 2e-5 * scale + 1e-12, (k, float((g1[k] - g0[k]).abs().max()), scale)
