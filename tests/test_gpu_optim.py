"""GPU parity of FusedAdam (csrc/adam.hip) against torch.optim.Adam run on CPU with the reference's optimizer setup
(lr=0.0 default, eps=1e-15, one lr per named group).  Tolerance 2e-6 relative per step on parameters and moments
(same float32 formula; only fused-multiply-add contraction differs)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _groups(tensors, lrs):
    return [{"params": [t], "lr": lr, "name": f"g{i}"} for i, (t, lr) in enumerate(zip(tensors, lrs))]


def _close(a, b, rel=2e-6):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert float((a - b).abs().max()) <= rel * float(b.abs().max()) + 1e-12, float((a - b).abs().max())


def test_matches_torch_adam_over_steps_with_reference_setup():
    from games_hip.optim import FusedAdam
    g = torch.Generator().manual_seed(0)
    shapes = [(1000, 3), (333, 1, 3), (4097,), (5, 15, 3), (1, 1), (70000, 3)]
    lrs = [1.6e-4, 1e-3, 2.5e-3, 1.25e-4, 0.05, 5e-3]
    cpu = [torch.randn(s, generator=g).requires_grad_(True) for s in shapes]
    gpu = [t.detach().clone().cuda().requires_grad_(True) for t in cpu]
    ref = torch.optim.Adam(_groups(cpu, lrs), lr=0.0, eps=1e-15)
    opt = FusedAdam(_groups(gpu, lrs), lr=0.0, eps=1e-15)
    for step in range(25):
        for a, b in zip(cpu, gpu):
            grad = torch.randn(a.shape, generator=g) * (10.0 ** float(torch.randint(-6, 2, (1,), generator=g)))
            if step == 3:
                grad[...] = 0                                   # zero gradient: denom = eps-dominated
            a.grad, b.grad = grad, grad.cuda()
        if step == 5:
            cpu[2].grad = None; gpu[2].grad = None              # a parameter without gradient is skipped
        ref.step(); opt.step()
        for a, b in zip(cpu, gpu):
            _close(b, a)
    for a, b in zip(cpu, gpu):
        _close(opt.state[b]["exp_avg"], ref.state[a]["exp_avg"])
        _close(opt.state[b]["exp_avg_sq"], ref.state[a]["exp_avg_sq"])
        assert float(opt.state[b]["step"]) == float(ref.state[a]["step"])


def test_more_tensors_than_one_launch_and_state_dict_roundtrip():
    from games_hip.optim import FusedAdam
    g = torch.Generator().manual_seed(1)
    cpu = [torch.randn(37 + i, generator=g).requires_grad_(True) for i in range(40)]
    gpu = [t.detach().clone().cuda().requires_grad_(True) for t in cpu]
    ref = torch.optim.Adam([{"params": cpu, "lr": 1e-2}], lr=0.0, eps=1e-15)
    opt = FusedAdam([{"params": gpu, "lr": 1e-2}], lr=0.0, eps=1e-15)
    for _ in range(3):
        for a, b in zip(cpu, gpu):
            a.grad = torch.randn(a.shape, generator=g); b.grad = a.grad.cuda()
        ref.step(); opt.step()
    sd = opt.state_dict()
    opt2 = FusedAdam([{"params": gpu, "lr": 1e-2}], lr=0.0, eps=1e-15)
    opt2.load_state_dict(sd)
    for a, b in zip(cpu, gpu):
        a.grad = torch.randn(a.shape, generator=g); b.grad = a.grad.cuda()
    ref.step(); opt2.step()
    for a, b in zip(cpu, gpu):
        _close(b, a)


def test_reference_optimizer_surgery_keeps_working():
    """scene/gaussian_model.py:284-300 (replace_tensor_to_optimizer): state is re-keyed to a new Parameter."""
    from games_hip.optim import FusedAdam
    p = torch.randn(64, 3).cuda().requires_grad_(True)
    opt = FusedAdam([{"params": [p], "lr": 1e-2, "name": "opacity"}], lr=0.0, eps=1e-15)
    p.grad = torch.randn_like(p); opt.step()
    group = opt.param_groups[0]
    stored = opt.state.get(group["params"][0], None)
    stored["exp_avg"] = torch.zeros_like(p); stored["exp_avg_sq"] = torch.zeros_like(p)
    del opt.state[group["params"][0]]
    newp = torch.nn.Parameter(torch.ones_like(p).requires_grad_(True))
    group["params"][0] = newp
    opt.state[newp] = stored
    newp.grad = torch.full_like(newp, 0.5); opt.step()
    assert torch.isfinite(newp).all() and float((newp - 1).abs().max()) > 0


def test_rejects_unsupported():
    from games_hip.optim import FusedAdam
    with pytest.raises(NotImplementedError):
        FusedAdam([torch.zeros(3, device="cuda", requires_grad=True)], weight_decay=0.1)
    p = torch.zeros(3, requires_grad=True)
    opt = FusedAdam([p], lr=1e-3)
    p.grad = torch.ones(3)
    with pytest.raises(RuntimeError):
        opt.step()
