"""BASELINE config 4 (gs_multi_mesh 'ficus', ~300k Gaussians, 800x800, views sharded one per GPU, gradient all-reduce) on
the GPU with REAL gradients:

  * one full-size view of `c4_ficus_like` (three meshes, 3 / 5 / 2 splats per face, 299 472 Gaussians) through
    HipGaussianMultiMeshModel against the oracle chain (torch-CPU multi-mesh K0 restatement + C rasterizer) under the suite's
    gradient criterion -- reference: games/multi_mesh_splatting/scene/gaussian_multi_mesh_model.py:99-199;
  * SURVEY.md 8(e) "Semantics caveat": two ranks (sharing cuda:0 over gloo on this 1-GPU box), rank r renders view r; after the
    exchange every parameter gradient equals the SUM of the two single-view gradients computed serially in one process -- for
    the ring, direct and direct+all-gather all-reduce, for the factorised SH exchange and for the packed single-collective exchange."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import _util as U
from games_hip import synthetic as syn

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_chain(scenes, okw, gc, dtype=torch.float32, precision="f32"):
    """parameters -> multi-mesh K0 (per-mesh restatement, concatenated) -> activations -> C rasterizer; returns (oracle out, grads)"""
    from oracle import gs_oracle, mesh_oracle
    d = lambda t: t.detach().to(dtype).clone().requires_grad_(True)
    vs, als, scs = [d(s.vertices) for s in scenes], [d(s._alpha) for s in scenes], [d(s._scale) for s in scenes]
    op, fdc, frest = d(torch.cat([s._opacity for s in scenes])), d(torch.cat([s._features_dc for s in scenes])), d(torch.cat([s._features_rest for s in scenes]))
    parts = [mesh_oracle.mesh_to_gaussians(v, s.faces, a, sc)[2:] for v, a, sc, s in zip(vs, als, scs, scenes)]
    xyz, scaling, rot = (torch.cat([p[i] for p in parts]) for i in range(3))
    xa, sa, ra, oa, shs = mesh_oracle.activated(xyz, scaling, rot, op, fdc, frest)
    o = gs_oracle.rasterize(means3D=xa, opacities=oa, shs=shs, scales=sa, rotations=ra, precision=precision, **okw)
    g = gs_oracle.backward(o, gc.to(dtype) if gc is not None else syn.upstream_grad(torch.from_numpy(o.color)) * 1000.0)
    ((xa * torch.from_numpy(g["means3D"])).sum() + (sa * torch.from_numpy(g["scales"])).sum() + (ra * torch.from_numpy(g["rotations"])).sum()
     + (oa * torch.from_numpy(g["opacities"])).sum() + (shs * torch.from_numpy(g["sh"])).sum()).backward()
    grads = {}
    for i in range(len(scenes)):
        grads[f"vertices{i}"], grads[f"_alpha{i}"], grads[f"_scale{i}"] = vs[i].grad.numpy(), als[i].grad.numpy(), scs[i].grad.numpy()
    grads.update(_opacity=op.grad.numpy(), f_dc=fdc.grad.numpy(), f_rest=frest.grad.numpy())
    return o, grads


def _model_grads(model):
    g = {}
    for i in range(len(model.vertices)):
        g[f"vertices{i}"], g[f"_alpha{i}"], g[f"_scale{i}"] = model.vertices[i].grad, model._alpha[i].grad, model._scale[i].grad
    g.update(_opacity=model._opacity.grad, f_dc=model._features_dc.grad, f_rest=model._features_rest.grad)
    return {k: v.detach().cpu().numpy() for k, v in g.items()}


def test_config4_full_size_view_matches_the_oracle_chain():
    from games_hip.model import HipGaussianMultiMeshModel
    from games_hip.render import PipelineParams, render
    scenes = syn.multi_mesh_scenes("c4_ficus_like", state="trained")
    size = scenes[0].meta["image"]
    cam = syn.orbit_camera(5, width=size, height=size)
    okw = {k: v for k, v in U.settings_kwargs(cam, torch.ones(3)).items() if k not in ("prefiltered", "debug")}
    o, go = _oracle_chain(scenes, okw, None)
    gc = syn.upstream_grad(torch.from_numpy(o.color)) * 1000.0
    model = HipGaussianMultiMeshModel.from_scenes(scenes, "cuda")
    assert model.get_xyz.shape[0] == 299472

    def run_hip():
        for p_ in model.parameters():
            p_.grad = None
        model.update_alpha(); model.prepare_scaling_rot()
        pkg_ = render(cam.to("cuda"), model, PipelineParams(), torch.ones(3, device="cuda"))
        (pkg_["render"] * gc.cuda()).sum().backward()
        return dict(pkg=pkg_, grads=_model_grads(model))
    # deterministic mode under the strict criterion first, then the float-atomics mode under the default one
    hres, _ = U.assert_grads_both_modes(run_hip, go, lambda: _oracle_chain(scenes, okw, gc, torch.float64, "f64")[1],
                                        where="c4_ficus_like 800x800 view 5 through the multi-mesh K0")
    pkg = hres["pkg"]
    h = dict(color=pkg["render"].detach().cpu().numpy(), radii=pkg["radii"].cpu().numpy(), invdepth=pkg["depth"].detach().cpu().numpy())
    ora = dict(color=o.color, radii=o.radii, invdepth=o.invdepth, details=o.state.details())
    rep = U.forward_report(h, ora, size, size, input_rounding=True)          # each side ran its own float32 K0 stage
    assert rep["radii_unexplained"] == 0 and rep["max_clean"] <= 1e-4 and rep["amb_frac"] < 0.02 and rep["psnr"] > 60.0, rep


@pytest.mark.parametrize("workload", ["multi_tiny", "c4_ficus_like"])
def test_config4_exchanged_gradients_equal_the_sum_of_the_single_view_gradients(tmp_path, workload):
    from games_hip.model import HipGaussianMultiMeshModel
    from games_hip.render import PipelineParams, render
    scenes = syn.multi_mesh_scenes(workload, state="trained")
    size = scenes[0].meta["image"]
    world = 2
    model = HipGaussianMultiMeshModel.from_scenes(scenes, "cuda")
    params = model.parameters()
    bg = torch.ones(3, device="cuda")
    # ---- serial: the two views one after the other in THIS process; upstream gradients fixed up front and shared with the ranks
    gcs, serial = [], None
    for r in range(world):
        cam = syn.orbit_camera(r, width=size, height=size).to("cuda")
        with torch.no_grad():
            model.update_alpha(); model.prepare_scaling_rot()
            img = render(cam, model, PipelineParams(), bg)["render"]
        gcs.append((syn.upstream_grad(img.cpu()) * 1000.0 / world).contiguous())
    torch.save(gcs, tmp_path / "upstream.pt")
    for r in range(world):
        for p in params:
            p.grad = None
        cam = syn.orbit_camera(r, width=size, height=size).to("cuda")
        model.update_alpha(); model.prepare_scaling_rot()
        img = render(cam, model, PipelineParams(), bg)["render"]
        (img * gcs[r].cuda()).sum().backward()
        g = [p.grad.detach().double().cpu() for p in params]
        serial = g if serial is None else [a + b for a, b in zip(serial, g)]
    # ---- two ranks
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "ddp.pt"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tests", "_c4_ddp_worker.py"), workload, str(out)],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    res = torch.load(out)
    ran = []
    for variant, grads in res.items():
        if isinstance(grads, str):              # gloo cannot run this collective on device tensors: reported by the worker
            assert variant in ("direct", "direct_ag"), (variant, grads)
            continue
        ran.append(variant)
        for k, (a, b) in enumerate(zip(grads, serial)):
            a, b = a.double().numpy(), b.numpy()
            scale = np.abs(b).max()
            # float atomics: two runs of ONE view already differ in the last bits, more on ill-conditioned rows (measured on the
            # MI355X: 0.999 quantile 3e-7 of the tensor's scale, worst entry 1.6e-4 relative); the exchange itself adds one
            # float32 rounding per element.  Bound: 1e-5 of the tensor's scale at the 0.999 quantile, and the north-star 1e-3
            # relative (floor: 1 % of the scale) on EVERY entry
            rel = np.abs(a - b) / (np.abs(b) + 1e-2 * scale + 1e-30)
            assert np.quantile(np.abs(a - b), 0.999) <= 1e-5 * scale + 1e-30 and rel.max() <= 1e-3, (workload, variant, k, float(rel.max()))
    assert "ring" in ran and "factor" in ran and "packed" in ran, res.keys()
