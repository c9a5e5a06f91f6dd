"""GPU: the animated render loop as a replayed hipGraph (games_hip.animate.GraphedAnimation; SURVEY.md section 7 step 9,
scripts/render_time_animated.py:68-87).  Inside the capture the rasterizer runs in its launches-only form
(GmsRasterForwardArgs.no_host_wait): no host wait for the instance count.  Replayed frames must equal the eager ones bit for bit
(the forward has no float atomics), the frame's counts must be readable from the device, and a frame that outgrows the captured
capacity must be detected and redone."""
import pytest
import torch

from games_hip import synthetic as syn

pytestmark = pytest.mark.gpu


def _deform(v, k):
    out = v.clone()
    out[:, 2] += 0.03 * k * torch.sin(3.0 * v[:, 0] + 0.5 * k)
    return out


def test_graphed_animation_replays_the_eager_frames_bit_for_bit():
    import diff_gaussian_rasterization as dgr
    from games_hip.animate import GraphedAnimation
    from games_hip.model import HipGaussianMeshModel
    from games_hip.render import PipelineParams, render_animated
    model = HipGaussianMeshModel.from_scene(syn.mesh_scene("small"), "cuda")
    view = syn.orbit_camera(2, width=128, height=128).to("cuda")
    bg = torch.ones(3, device="cuda")
    pipe = PipelineParams()
    faces = model.faces.long()
    rest = model.vertices.detach().clone()
    anim = GraphedAnimation(model, view, pipe, bg)
    with torch.no_grad():
        for k in range(5):
            tri = _deform(rest, k)[faces].float()
            got = anim.render(tri, check=True).clone()
            st = anim.status()
            want = render_animated(None, tri, view, model, pipe, bg)["render"]
            assert torch.equal(got, want), k
            assert st["complete"] and st["num_rendered"] == dgr.last_stats()["num_rendered"], (k, st)
    assert anim.captures == 1                                   # five frames, one capture: the replays did the work
    # a frame that outgrows the capture: captured on the mesh shrunk to a tenth (5.5 k instances at 256x256 -> capacity ~11 k), then
    # the rest pose (24 k instances)
    view2 = syn.orbit_camera(2, width=256, height=256).to("cuda")
    with torch.no_grad():
        small_tri = (rest * 0.1)[faces].float()
        big = rest[faces].float()
        want = render_animated(None, big, view2, model, pipe, bg)["render"].clone()
        n_big = dgr.last_stats()["num_rendered"]
        dgr.clear_capacity_hints()
        anim2 = GraphedAnimation(model, view2, pipe, bg)
        anim2.render(small_tri)
        assert anim2.captures == 1 and anim2.status()["complete"] and n_big > anim2.capacity, (n_big, anim2.capacity)
        anim2.render(big, check=False)                          # without the check: the overflow shows in the frame's counts
        st = anim2.status()
        assert not st["complete"] and st["num_rendered"] == n_big, st
        got = anim2.render(big, check=True).clone()             # with it: re-captured at 1.5x the frame's count and redone
        assert anim2.captures == 2 and anim2.status()["complete"]
        assert torch.equal(got, want)


def test_capture_without_a_warmed_up_shape_is_refused():
    """The launches-only forward needs the capacity hint of its shape: capturing a cold shape raises instead of sizing blindly."""
    import diff_gaussian_rasterization as dgr
    from games_hip.model import HipGaussianMeshModel
    from games_hip.render import PipelineParams, render
    dgr.clear_capacity_hints()
    model = HipGaussianMeshModel.from_scene(syn.mesh_scene("tiny"), "cuda")
    view = syn.orbit_camera(1, width=80, height=72).to("cuda")      # a shape no other test warms up
    bg = torch.ones(3, device="cuda")
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.stream(s):
        with pytest.raises(Exception):
            with torch.cuda.graph(g, stream=s):
                render(view, model, PipelineParams(), bg)
