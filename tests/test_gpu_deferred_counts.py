"""GPU: deferred read-back of the frame's instance count (GmsRasterForwardArgs.count_ticket_out + gms_rasterize_forward_counts, ABI 7;
diff_gaussian_rasterization.set_deferred_counts).  The blocking form waits for N inside the forward (what the upstream binding's
`num_rendered` read-back does); the deferred form enqueues, hands out a ticket and reads N at the start of the
backward.  Same frame, same gradients; an overflowed frame is REPORTED by the backward (RuntimeError carrying DEFERRED_OVERFLOW) and a
redone step is correct."""
import pytest
import torch

from games_hip import synthetic as syn

pytestmark = pytest.mark.gpu
PARAMS = ("vertices", "_alpha", "_scale", "_opacity", "_features_dc", "_features_rest")


@pytest.fixture
def dgr_state():
    import diff_gaussian_rasterization as dgr
    was_det, was_def = dgr.deterministic(), dgr.deferred_counts()
    yield dgr
    dgr.set_deterministic(was_det)
    dgr.set_deferred_counts(was_def)


def _model(name="small"):
    from games_hip.model import HipGaussianMeshModel
    return HipGaussianMeshModel.from_scene(syn.mesh_scene(name), "cuda")


def _step(model, cam, bg, defer_k0=False):
    from games_hip.render import PipelineParams, render
    model.hip_defer_k0 = defer_k0
    for n in PARAMS:
        getattr(model, n).grad = None
    model.update_alpha(); model.prepare_scaling_rot()
    out = render(cam, model, PipelineParams(), bg)
    img = out["render"]
    (img * ((img.detach() - 0.5) / img.numel() * 1000.0)).sum().backward()
    return img.detach().clone(), {n: getattr(model, n).grad.detach().clone() for n in PARAMS}


@pytest.mark.parametrize("defer_k0", [False, True])
def test_deferred_and_blocking_forms_give_the_same_frame_and_gradients(dgr_state, defer_k0):
    dgr = dgr_state
    model = _model()
    cam = syn.orbit_camera(2, width=160, height=144).to("cuda")
    bg = torch.tensor([0.9, 0.7, 0.3], device="cuda")
    dgr.set_deterministic(True)                              # fixed summation order: the two forms must agree bit for bit
    dgr.set_deterministic(False); dgr.set_deferred_counts(False)
    _step(model, cam, bg); _step(model, cam, bg)            # (the first frames of a shape size their buffers with a blocking read-back)
    ref_img, ref = _step(model, cam, bg, defer_k0)
    dgr.set_deferred_counts(True)
    img, got = _step(model, cam, bg, defer_k0)
    stats = dgr.last_stats()
    assert stats["num_rendered"] > 0                         # redeemed by the backward
    model.hip_defer_k0 = False
    assert torch.equal(img, ref_img)
    for k in ref:
        scale = float(ref[k].abs().max())
        assert float((got[k] - ref[k]).abs().max()) <= 2e-5 * scale + 1e-12, k          # (float atomics: order-of-summation noise only)
    # seventeen forwards without a backward in between: the ring holds sixteen tickets, the oldest one has expired
    from games_hip.render import PipelineParams, render
    outs = []
    for _ in range(17):
        model.update_alpha(); model.prepare_scaling_rot()
        outs.append(render(cam, model, PipelineParams(), bg)["render"])
    with pytest.raises(RuntimeError, match="ticket expired"):
        outs[0].sum().backward()
    outs[-1].sum().backward()                                # the newest ticket is fine


def test_an_overflowed_deferred_frame_is_reported_by_the_backward_and_the_redone_step_is_right(dgr_state):
    dgr = dgr_state
    from games_hip.render import PipelineParams, render
    model = _model()
    cam = syn.orbit_camera(1, width=128, height=128).to("cuda")
    bg = torch.ones(3, device="cuda")
    dgr.set_deferred_counts(False)
    ref_img, ref = _step(model, cam, bg)
    n_true = dgr.last_stats()["num_rendered"]
    P = int(model._scale.numel())
    dgr.set_deferred_counts(True)
    dgr.set_capacity_hint(torch.cuda.current_device(), 128, 128, P, max(1, n_true // 8))        # "the scene has grown 8x since the last frame"
    for n in PARAMS:
        getattr(model, n).grad = None
    model.update_alpha(); model.prepare_scaling_rot()
    img = render(cam, model, PipelineParams(), bg)["render"]
    with pytest.raises(RuntimeError, match=dgr.DEFERRED_OVERFLOW):
        (img * ((img.detach() - 0.5) / img.numel() * 1000.0)).sum().backward()
    # the capacity hint has been raised by the redeemed count: the redone step is complete and equals the blocking one
    img2, got = _step(model, cam, bg)
    assert torch.equal(img2, ref_img)
    for k in ref:
        scale = float(ref[k].abs().max())
        assert float((got[k] - ref[k]).abs().max()) <= 2e-5 * scale + 1e-12, k


def test_training_loop_redoes_an_overflowed_step(dgr_state, monkeypatch):
    """games_hip.train.training with the deferred read-back: a frame that overflows mid-run is redone, the run ends finite and improved."""
    dgr = dgr_state
    import random
    from games_hip.render import PipelineParams, render
    from games_hip.train import OptimizationParamsMesh, training
    random.seed(0); torch.manual_seed(0)
    model = _model()
    opt = OptimizationParamsMesh(iterations=12, vertices_lr=0.00016)
    model.training_setup(vertices_lr=opt.vertices_lr, alpha_lr=opt.alpha_lr, feature_lr=opt.feature_lr, opacity_lr=opt.opacity_lr,
                         scaling_lr=opt.scaling_lr, fused=True)
    cams = []
    for k in range(3):
        c = syn.orbit_camera(k, width=96, height=96).to("cuda")
        c.original_image = torch.rand(3, 96, 96, device="cuda", generator=torch.Generator(device="cuda").manual_seed(k))
        cams.append(c)
    model.update_alpha(); model.prepare_scaling_rot()
    dgr.set_deferred_counts(True)
    P = int(model._scale.numel())
    calls = [0]

    def render_with_a_shrunk_hint(cam, pc, pipe, bg_):
        calls[0] += 1
        if calls[0] == 6:                                   # once, mid-run: pretend the learnt instance count is far too small
            dgr.set_capacity_hint(torch.cuda.current_device(), 96, 96, P, 64)
        return render(cam, pc, pipe, bg_)

    losses = training(model, cams, opt, PipelineParams(), torch.ones(3, device="cuda"), render=render_with_a_shrunk_hint, report_iterations=[1, 12])
    assert calls[0] == 13                                   # twelve iterations + the one redone step
    assert all(torch.isfinite(getattr(model, n)).all() for n in PARAMS) and losses[1] < losses[0]
