"""GPU: forward-only animated frames rendered straight from the mesh (GmsRasterForwardArgs.mesh, ABI 5; SURVEY.md section 7 step 9,
scripts/render_time_animated.py:68-87).  The preprocess thread derives its Gaussian from the face (the arithmetic of the K0 kernel),
so there is no K0 launch and no xyz / scale / rotation tensor; the frame must equal the unfused one -- K0 launch, then the
rasterizer -- bit for bit (the forward has no float atomics), for explicit triangles, for vertices + faces, and inside a replayed
hipGraph."""
import pytest
import torch

from games_hip import synthetic as syn

pytestmark = pytest.mark.gpu


def _deform(v, k):
    out = v.clone()
    out[:, 2] += 0.03 * k * torch.sin(3.0 * v[:, 0] + 0.5 * k)
    return out


def _model(name="small"):
    from games_hip.model import HipGaussianMeshModel
    return HipGaussianMeshModel.from_scene(syn.mesh_scene(name), "cuda")


@pytest.mark.parametrize("size", [128, 203])
def test_fused_frame_equals_the_k0_launch_plus_rasterizer_bit_for_bit(monkeypatch, size):
    import diff_gaussian_rasterization as dgr
    from games_hip.animate import render_frame
    from games_hip.render import PipelineParams, _fused_frame_ok, render_animated
    model = _model()
    view = syn.orbit_camera(2, width=size, height=size - 5).to("cuda")
    bg = torch.tensor([0.9, 0.7, 0.3], device="cuda")
    pipe = PipelineParams()
    faces = model.faces.long()
    rest = model.vertices.detach().clone()
    with torch.no_grad():
        assert _fused_frame_ok(model, pipe, None)
        for k in range(4):
            v = _deform(rest, k)
            monkeypatch.setenv("GMS_ANIMATE_FUSED", "0")
            want = render_animated(None, v[faces].float(), view, model, pipe, bg)
            n_want = dgr.last_stats()["num_rendered"]
            monkeypatch.setenv("GMS_ANIMATE_FUSED", "1")
            got_tri = render_animated(None, v[faces].float(), view, model, pipe, bg)          # explicit triangles (the reference's call)
            got_mesh = render_frame(v, faces, view, model, pipe, bg)                           # vertices + faces: no gather either
            assert dgr.last_stats()["num_rendered"] == n_want
            for got in (got_tri, got_mesh):
                assert torch.equal(got["render"], want["render"]), k
                assert torch.equal(got["radii"], want["radii"]) and torch.equal(got["depth"], want["depth"]), k
                assert torch.equal(got["visibility_filter"], want["visibility_filter"]), k
        assert want["render"].std().item() > 0.01


def test_fused_frame_is_not_taken_where_the_frame_is_differentiated_or_the_model_does_not_fit():
    from games_hip.render import PipelineParams, _fused_frame_ok, render_animated
    model = _model()
    pipe = PipelineParams()
    assert not _fused_frame_ok(model, pipe, None)                        # grad mode on: the frame may be differentiated
    view = syn.orbit_camera(1, width=96, height=96).to("cuda")
    bg = torch.ones(3, device="cuda")
    tri = model.vertices[model.faces.long()].float()
    out = render_animated(None, tri, view, model, pipe, bg)             # ... and takes the K0 launch + rasterizer, as before
    assert out["viewspace_points"] is not None and out["render"].requires_grad
    with torch.no_grad():
        assert _fused_frame_ok(model, pipe, None)
        assert not _fused_frame_ok(model, PipelineParams(convert_SHs_python=True), None)
        assert not _fused_frame_ok(model, pipe, torch.zeros(3, device="cuda"))
        model.active_sh_degree = 2                                       # the SH ramp of training: lower degrees keep the K0 launch
        assert not _fused_frame_ok(model, pipe, None)
        model.active_sh_degree = 3
        model._opacity.add_(0.25)                                        # the cached kernel sigmoid is stale now: get_opacity would
        assert not _fused_frame_ok(model, pipe, None)                    # fall back to torch.sigmoid, so the fused path steps aside
        model.update_alpha(); model.prepare_scaling_rot()
        assert _fused_frame_ok(model, pipe, None)


def test_c_abi_rejects_an_incomplete_fused_request():
    """The C ABI itself: mesh->P must equal P, split degree-3 storage is required."""
    import ctypes as C
    from diff_gaussian_rasterization import _lib
    lib = _lib.load()
    a = _lib.RasterForwardArgs()
    m = (C.c_byte * 256)()                                              # a zeroed GmsMeshArgs: P = 0, null pointers
    a.P, a.D, a.M, a.width, a.height = 10, 3, 16, 64, 64
    buf = torch.zeros(4096, device="cuda")
    a.out_color = a.out_invdepth = a.background = buf.data_ptr()
    a.mesh = C.cast(m, C.c_void_p).value
    rc = lib.gms_rasterize_forward(C.byref(a), None)
    assert rc == -1 and b"fused mesh input" in lib.gms_last_error()


def test_fused_frame_rejects_a_mesh_that_does_not_match_the_model():
    """F * splats_per_face must equal P before the preprocess thread indexes faces / vertices -- through the
    python driver, through the torch binding and through the C ABI itself."""
    import ctypes as C
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _lib
    from games_hip.render import PipelineParams, render_animated, render_mesh_frame
    model = _model()
    view = syn.orbit_camera(1, width=96, height=96).to("cuda")
    bg = torch.ones(3, device="cuda")
    pipe = PipelineParams()
    faces = model.faces.long()
    with torch.no_grad():
        tri = model.vertices[faces].float()
        with pytest.raises(RuntimeError, match="triangles for a model"):
            render_animated(None, tri[:-3], view, model, pipe, bg)
        with pytest.raises(RuntimeError, match="splats per face"):
            render_mesh_frame(model.vertices, faces[:-3], view, model, pipe, bg)
        with pytest.raises(RuntimeError, match=r"\(num_faces, 3\)"):
            render_mesh_frame(model.vertices, faces.reshape(-1, 1), view, model, pipe, bg)
        assert torch.isfinite(render_mesh_frame(model.vertices, faces, view, model, pipe, bg)["render"]).all()
    # the C ABI: a GmsMeshArgs whose F * splats_per_face != P is refused before any launch
    lib = _lib.load()
    a = _lib.RasterForwardArgs()
    m = _lib.MeshArgs()
    if True:
        P = int(model._scale.numel())
        m.F, m.V, m.P, m.splats_per_face = int(faces.shape[0]) - 1, int(model.vertices.shape[0]), P, int(model._alpha.shape[1])
        buf = torch.zeros(4096, device="cuda")
        m.vertices = m._alpha = m._scale = m._opacity = buf.data_ptr(); m.faces = faces.data_ptr()
        a.P, a.D, a.M, a.width, a.height = P, 3, 16, 64, 64
        a.out_color = a.out_invdepth = a.background = buf.data_ptr()
        a.mesh = C.cast(C.pointer(m), C.c_void_p).value
        rc = lib.gms_rasterize_forward(C.byref(a), None)
        assert rc == -1 and b"P != F * splats_per_face" in lib.gms_last_error()


def test_graphed_animation_replays_fused_frames():
    from games_hip.animate import GraphedAnimation
    from games_hip.render import PipelineParams, render_animated
    import os
    model = _model()
    view = syn.orbit_camera(3, width=160, height=160).to("cuda")
    bg = torch.ones(3, device="cuda")
    pipe = PipelineParams()
    faces = model.faces.long()
    rest = model.vertices.detach().clone()
    anim = GraphedAnimation(model, view, pipe, bg)
    with torch.no_grad():
        for k in range(4):
            tri = _deform(rest, k)[faces].float()
            got = anim.render(tri).clone()
            os.environ["GMS_ANIMATE_FUSED"] = "0"
            try:
                want = render_animated(None, tri, view, model, pipe, bg)["render"]
            finally:
                os.environ.pop("GMS_ANIMATE_FUSED", None)
            assert torch.equal(got, want), k
    assert anim.captures == 1
