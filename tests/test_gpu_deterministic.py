"""GPU: the deterministic-reduction mode (include/gmsplat.h, gms_set_deterministic / GAMES_HIP_DETERMINISTIC=1; SURVEY.md
section 5 "race detection", section 7 hard part 1 "keep a deterministic mode for tests").

Default mode accumulates gradients with float atomics, as the reference's CUDA rasterizer does: two runs
differ by ~1e-7..1e-6 relative.  With the mode on, every sum of the backward passes runs in a fixed order: the gradients of two
runs are BIT-IDENTICAL -- asserted here for every parameter of the mesh model at the headline size (micro-tile kernels), at the
config-5 size (quadrant kernels, 11.8 M instances) and through both compositing implementations on a small scene -- and the
parity criterion is applied WITHOUT its two atomics-order allowances (`assert_grads(strict=True)`)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import _util as U
from games_hip import synthetic as syn

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_RUNS = r'''
import sys, os, json, torch, numpy as np
sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import conftest
import diff_gaussian_rasterization as dgr
from games_hip import synthetic as syn
from games_hip.model import HipGaussianMeshModel, HipGaussianFlameModel
from games_hip.render import PipelineParams, render
assert dgr.deterministic() == %(det)s
scene = syn.mesh_scene(%(scene)r)
model = (HipGaussianFlameModel if %(flame)s else HipGaussianMeshModel).from_scene(scene, 'cuda')
model.active_sh_degree = 3
cam = syn.orbit_camera(%(view)d, width=scene.meta["image"], height=scene.meta["image"]).to("cuda")
bg = torch.ones(3, device='cuda')
params = model.parameters()
def run():
    for p in params:
        p.grad = None
    model.update_alpha(); model.prepare_scaling_rot()
    out = render(cam, model, PipelineParams(), bg)
    img = out['render']
    img.backward(syn.upstream_grad(img.detach()) * 1000.0)
    torch.cuda.synchronize()
    return [p.grad.detach().clone() for p in params if p.grad is not None] + [out['viewspace_points'].grad.detach().clone(), img.detach().clone()]
runs = [run() for _ in range(3)]
same = all(torch.equal(a, b) for r in runs[1:] for a, b in zip(runs[0], r))
maxrel = max(float((a - b).abs().max() / (a.abs().max() + 1e-30)) for r in runs[1:] for a, b in zip(runs[0], r))
nz = all(float(a.abs().max()) > 0 for a in runs[0])
print('RESULT', json.dumps(dict(same=bool(same), maxrel=maxrel, nonzero=bool(nz), n=len(runs[0]), P=int(model.get_xyz.shape[0]))))
'''


def _three_runs(scene, det, env=None, flame=False, view=1):
    code = _RUNS % dict(root=ROOT, det=1 if det else 0, scene=scene, flame=bool(flame), view=view)
    e = dict(os.environ, **(env or {}))
    e.pop("GAMES_HIP_DETERMINISTIC", None)
    if det:
        e["GAMES_HIP_DETERMINISTIC"] = "1"
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and "RESULT" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
    import json
    return json.loads(r.stdout.split("RESULT", 1)[1])


def test_bit_identical_gradients_at_the_headline_size():
    """c2_hotdog_like (299 712 mesh-bound Gaussians, 800x800, K0 -> render -> backward -> K0 backward): three runs, every
    parameter gradient (vertices, _alpha, features, opacity, scale), the screen-space gradient and the image are bit-equal."""
    res = _three_runs("c2_hotdog_like", det=True)
    assert res["same"] and res["nonzero"] and res["P"] == 299712 and res["n"] >= 7, res


def test_default_mode_is_order_dependent_but_within_the_atomics_noise():
    """The control: the same three runs with float atomics agree to ~1e-6 of each tensor's scale and are,
    at this size, not bit-identical -- which is what the mode above removes."""
    res = _three_runs("c2_hotdog_like", det=False)
    assert res["maxrel"] < 2e-5 and res["nonzero"], res
    assert not res["same"], "float-atomics runs came out bit-identical: the control no longer shows what the mode is for"


def test_bit_identical_gradients_at_config5_size_quadrant_kernels():
    """997 600 Gaussians, 1024x1024, 11.8 M instances, tiles 26 k deep: the quadrant-wave kernels with per-(instance, quadrant)
    partial records, the flame model's softmax alphas and one-wave-per-face K0 backward."""
    res = _three_runs("c5_flame_like_1m", det=True, flame=True)
    assert res["same"] and res["nonzero"] and res["P"] == 997600, res


@pytest.mark.parametrize("env", [{"GMS_MICRO": "1"}, {"GMS_MICRO": "0"}])
def test_both_compositing_implementations_are_deterministic_on_a_small_scene(env):
    res = _three_runs("small", det=True, env=env)
    assert res["same"] and res["nonzero"], res


@pytest.mark.parametrize("case", ["random", "flat10k"])
def test_parity_criterion_without_the_atomics_order_allowances(case):
    """In deterministic mode `assert_grads(strict=True)`: K = 4 instead of 8 and not one unexplained entry (instead of one per million).  Forward parity is
    unchanged by the mode (the forward has no float atomics)."""
    import diff_gaussian_rasterization as dgr
    if case == "random":
        sc, cam = syn.random_scene(4000, seed=1, scale_lo=0.01, scale_hi=0.1), syn.orbit_camera(1, width=160, height=128, radius=3.0)
        bg = torch.tensor([0.2, 0.4, 0.6])
    else:
        sc, cam = syn.flat_scene(10000), syn.orbit_camera(0, width=256, height=256)
        bg = torch.ones(3)
    inputs = dict(means3D=sc.means3D, opacities=sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    kw = U.settings_kwargs(cam, bg)
    W, H = cam.image_width, cam.image_height
    o = U.oracle_render(inputs, kw)
    gc = syn.upstream_grad(torch.from_numpy(o["color"])).numpy() * 1000.0
    gdm = np.full((1, H, W), 1e-3, np.float32)
    o = U.oracle_render(inputs, kw, gc, gdm)
    was = dgr.deterministic()
    dgr.set_deterministic(True)
    try:
        h = U.hip_render(inputs, kw, grad_color=gc, grad_invdepth=gdm)
        h2 = U.hip_render(inputs, kw, grad_color=gc, grad_invdepth=gdm)
    finally:
        dgr.set_deterministic(was)
    for k, v in h["grads"].items():
        if v is not None:
            assert np.array_equal(v, h2["grads"][k]), k
    rep = U.forward_report(h, o, W, H)
    assert rep["radii_unexplained"] == 0 and rep["max_clean"] <= 1e-4, rep
    U.assert_grads(h["grads"], o["grads"], lambda: U.oracle_render(inputs, kw, gc, gdm, precision="f64")["grads"],
                   where=f"deterministic strict {case}", excuse=U.excused_rows(o["details"]),
                   go32acc_fn=lambda: U.f32_realisations(inputs, kw, gc, gdm), alt=U.alt_oracles(inputs, kw, gc, gdm, o["details"]), strict=True)
