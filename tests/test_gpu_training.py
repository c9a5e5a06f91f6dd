"""End-to-end sanity on the GPU: the reference's training iteration (train.py:90-157) on the HIP stack -- fused K0,
rasterizer, fused L1+SSIM loss, FusedAdam -- must actually fit a target."""
import pytest
import torch

from games_hip import synthetic as syn

pytestmark = pytest.mark.gpu


def test_training_loop_fits_a_target_image():
    from games_hip.loss import l1_ssim_loss
    from games_hip.model import HipGaussianMeshModel
    from games_hip.render import PipelineParams, render
    torch.manual_seed(0)
    scene = syn.mesh_scene("tiny")
    cams = [syn.orbit_camera(k, width=96, height=96).to("cuda") for k in (0, 2, 5)]
    bg = torch.ones(3, device="cuda")
    pipe = PipelineParams()
    teacher = HipGaussianMeshModel.from_scene(scene, "cuda")
    with torch.no_grad():
        targets = [render(c, teacher, pipe, bg)["render"].clone() for c in cams]
    student = HipGaussianMeshModel.from_scene(scene, "cuda")
    with torch.no_grad():                                   # perturb everything the optimizer owns
        student._features_dc.add_(0.4 * torch.randn_like(student._features_dc))
        student._features_rest.mul_(0.0)
        student._opacity.add_(-1.0)
        student._scale.mul_(0.7)
        student.vertices.add_(0.01 * torch.randn_like(student.vertices))
    student.training_setup(vertices_lr=1e-4, alpha_lr=1e-3, feature_lr=1e-2, opacity_lr=5e-2, scaling_lr=5e-3, fused=True)

    def total_loss():
        with torch.no_grad():
            student.update_alpha(); student.prepare_scaling_rot()
            return sum(float(l1_ssim_loss(render(c, student, pipe, bg)["render"], t, 0.2)) for c, t in zip(cams, targets))

    before = total_loss()
    for it in range(150):
        k = it % len(cams)
        student.update_alpha(); student.prepare_scaling_rot()
        out = render(cams[k], student, pipe, bg)
        loss = l1_ssim_loss(out["render"], targets[k], 0.2)
        loss.backward()
        assert out["viewspace_points"].grad is not None      # densification statistics input (train.py:129-133)
        student.optimizer.step()
        student.optimizer.zero_grad(set_to_none=True)
    after = total_loss()
    assert after < 0.5 * before, (before, after)
    for p in student.parameters():
        assert torch.isfinite(p).all()
