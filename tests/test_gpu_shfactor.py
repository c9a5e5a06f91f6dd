"""GPU tests of the factorised SH gradient (include/gmsplat.h: dL_dcolors on the SH path + gms_sh_grad_expand): the HIP path
against its own dense SH gradient and against the oracle's factor.  Two backward calls never agree bit for bit (float
atomics in blend_bwd), so HIP-vs-HIP comparisons carry a 1e-5 tolerance relative to the tensor's scale."""
import numpy as np
import pytest
import torch

import _util as U
from games_hip import synthetic as syn

pytestmark = pytest.mark.gpu


def _scene(P=4000, size=96, view=1, deg=3):
    sc = syn.random_scene(P, seed=11, scale_lo=0.02, scale_hi=0.3, opacity_lo=0.2, opacity_hi=0.9)
    cam = syn.orbit_camera(view, width=size, height=size - 16)
    kw = U.settings_kwargs(cam, torch.tensor([0.1, 0.3, 0.2]), sh_degree=deg)
    inputs = dict(means3D=sc.means3D, opacities=sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    return sc, cam, kw, inputs


def _close(a, b, rel=1e-5):
    return float((a - b).abs().max()) <= rel * float(b.abs().max()) + 1e-30


def _run(inputs, kw, gc, split=False):
    """forward + backward through the drop-in; returns (grads dict of torch tensors or None, tensors)."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, SplitSH
    dev = torch.device("cuda")
    t = {k: v.to(dev).float().detach().clone().requires_grad_(True) for k, v in inputs.items() if k != "shs"}
    shs = inputs["shs"].to(dev).float()
    if split:
        dc, rest = shs[:, :1].contiguous().requires_grad_(True), shs[:, 1:].contiguous().requires_grad_(True)
        sh_arg = SplitSH(dc, rest)
    else:
        full = shs.clone().requires_grad_(True)
        sh_arg = full
    kwd = dict(kw)
    for k in ("bg", "viewmatrix", "projmatrix", "campos"):
        kwd[k] = kwd[k].to(dev).float()
    means2D = torch.zeros_like(t["means3D"], requires_grad=True)
    color, radii, invd = GaussianRasterizer(GaussianRasterizationSettings(**kwd))(
        means3D=t["means3D"], means2D=means2D, opacities=t["opacities"], shs=sh_arg, scales=t["scales"], rotations=t["rotations"])
    (color * torch.as_tensor(gc, device=dev)).sum().backward()
    g = {k: v.grad for k, v in t.items()}
    g["means2D"] = means2D.grad
    if split:
        g["dc"], g["rest"] = dc.grad, rest.grad
    else:
        g["shs"] = full.grad
    return g, t


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("deg", [3, 1])
def test_one_view_factor_then_expand_equals_the_dense_sh_gradient(split, deg):
    import diff_gaussian_rasterization as dgr
    if dgr._C is None:
        pytest.skip("factorised mode needs the _C binding")
    sc, cam, kw, inputs = _scene(deg=deg)
    o = U.oracle_render(inputs, kw)
    gc = (syn.upstream_grad(torch.from_numpy(o["color"])).numpy() * 100.0).astype(np.float32)
    o = U.oracle_render(inputs, kw, gc, None)
    dense, _ = _run(inputs, kw, gc, split=split)
    dgr.set_sh_factor_mode(True)
    try:
        fac, t = _run(inputs, kw, gc, split=split)
        queued = dgr.take_sh_factors()
    finally:
        dgr.set_sh_factor_mode(False)
    assert len(queued) == 1 and tuple(queued[0].shape) == (sc.means3D.shape[0] + 1, 3)
    # no SH gradient from autograd in this mode; every other gradient is unchanged
    assert all(fac[k] is None for k in (("dc", "rest") if split else ("shs",)))
    for k in ("means3D", "means2D", "opacities", "scales", "rotations"):
        assert _close(fac[k], dense[k]), k
    P = sc.means3D.shape[0]
    assert torch.equal(queued[0][P].cpu(), kw["campos"].float().reshape(3))
    ref = torch.from_numpy(np.asarray(o["sh_factor"], np.float32))
    got = queued[0][:P].cpu()
    assert float((got - ref).abs().max()) <= 1e-3 * float(ref.abs().max()) + 1e-12          # (float atomics: tolerance of the suite)
    # expand == dense (same basis expressions, one product per coefficient; the factor itself is a second run's atomics)
    dev = queued[0].device
    if split:
        dc, rest = torch.full((P, 1, 3), 7.0, device=dev), torch.full((P, 15, 3), 7.0, device=dev)
        dgr.sh_grad_expand(queued[0][None].contiguous(), t["means3D"], deg, dc, rest)
        assert _close(dc, dense["dc"]) and _close(rest, dense["rest"])
        assert float(rest[:, (deg + 1) ** 2 - 1:].abs().max() if deg < 3 else 0.0) == 0.0          # above the active degree: exact zeros
    else:
        full = torch.full((P, 16, 3), 7.0, device=dev)
        dgr.sh_grad_expand(queued[0][None].contiguous(), t["means3D"], deg, full)
        assert _close(full, dense["shs"])
        assert float(full[:, (deg + 1) ** 2:].abs().max() if deg < 3 else 0.0) == 0.0
        acc = dense["shs"].clone()
        dgr.sh_grad_expand(queued[0][None].contiguous(), t["means3D"], deg, acc, accumulate=True)
        assert _close(acc, 2 * dense["shs"])


def test_two_views_expand_equals_the_sum_of_the_dense_gradients_and_the_exchange_sets_grads():
    import diff_gaussian_rasterization as dgr
    from games_hip.ddp import ShFactorExchange
    if dgr._C is None:
        pytest.skip("factorised mode needs the _C binding")
    dense_sum, facs, pos = None, [], None
    for view in (0, 5):
        sc, cam, kw, inputs = _scene(view=view)
        o = U.oracle_render(inputs, kw)
        gc = (syn.upstream_grad(torch.from_numpy(o["color"])).numpy() * 100.0).astype(np.float32)
        d, t = _run(inputs, kw, gc)
        dense_sum = d["shs"] if dense_sum is None else dense_sum + d["shs"]
        pos = t["means3D"]
    P = pos.shape[0]
    f_dc = torch.zeros(P, 1, 3, device="cuda", requires_grad=True)
    f_rest = torch.zeros(P, 15, 3, device="cuda", requires_grad=True)
    ex = ShFactorExchange(f_dc, f_rest, world=1).enable()
    try:
        for view in (0, 5):
            sc, cam, kw, inputs = _scene(view=view)
            o = U.oracle_render(inputs, kw)
            gc = (syn.upstream_grad(torch.from_numpy(o["color"])).numpy() * 100.0).astype(np.float32)
            _run(inputs, kw, gc)
        ex.start()
        ex.finish(pos, 3)
    finally:
        ex.disable()
    got = torch.cat([f_dc.grad, f_rest.grad], dim=1)
    scale = float(dense_sum.abs().max())
    assert float((got - dense_sum).abs().max()) <= 2e-6 * scale
    assert dgr.take_sh_factors() == [] and not dgr.sh_factor_mode()
