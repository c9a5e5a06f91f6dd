"""FusedAdam: `torch.optim.Adam` for the reference's training_setup() with the whole step in one HIP launch.

Drop-in for `torch.optim.Adam(l_params, lr=0.0, eps=1e-15)` (games/mesh_splatting/scene/gaussian_mesh_model.py:183,
scene/gaussian_model.py:160): same constructor, same `param_groups` (per-group 'lr' / 'name'), same `state` layout
({'step', 'exp_avg', 'exp_avg_sq'} per parameter) so the reference's optimizer surgery during densification
(scene/gaussian_model.py:284-375 replace/cat/prune of `optimizer.state`) and `state_dict()` checkpoints keep working.
csrc/adam.hip does the arithmetic of torch/optim/adam.py::_single_tensor_adam; GPU float32 parameters only."""
import ctypes as C

import torch

import diff_gaussian_rasterization as _dgr
from diff_gaussian_rasterization import _lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("FusedAdam implements what the reference uses: no weight decay, no amsgrad")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        by_hyper = {}
        keep, updated = [], []
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            lr, eps = float(group["lr"]), float(group["eps"])
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                state = self.state[p]
                if len(state) == 0:
                    if g.is_sparse:
                        raise RuntimeError("FusedAdam does not support sparse gradients")
                    _lib.require_gpu(p)
                    if p.dtype != torch.float32 or not p.is_contiguous():
                        raise RuntimeError("FusedAdam needs contiguous float32 parameters")
                    # `step` is a host-side count (a Python float, as torch.optim.Adam kept it before 1.12): the
                    # kernel needs it on the host for the bias corrections and a CPU tensor costs ~5 us per update
                    state["step"] = 0.0
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                step = float(state["step"]) + 1.0          # float(): also accepts a tensor from a torch.optim.Adam checkpoint
                state["step"] = step
                if g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.float().contiguous()
                    keep.append(g)
                m, v = state["exp_avg"], state["exp_avg_sq"]
                if not (m.is_contiguous() and v.is_contiguous()):
                    m, v = state["exp_avg"], state["exp_avg_sq"] = m.contiguous(), v.contiguous()
                key = (p.device, float(beta1), float(beta2), eps)
                lst = by_hyper.get(key)
                if lst is None:
                    lst = by_hyper[key] = []
                lst.append((p, g, m, v, lr, int(step)) if _dgr._C is not None else
                           _lib.AdamTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, int(step)))
                updated.append(p)
        for (device, beta1, beta2, eps), tensors in by_hyper.items():
            if _dgr._C is not None:
                ps, gs, ms, vs, lrs, steps = zip(*tensors)
                _dgr._C.adam_step(list(ps), list(gs), list(ms), list(vs), list(lrs), list(steps), beta1, beta2, eps)
                continue
            arr = (_lib.AdamTensor * len(tensors))(*tensors)
            with _lib.on_device(device):
                rc = lib.gms_adam_step(arr, len(tensors), beta1, beta2, eps, C.c_void_p(_lib.stream_ptr(device)))
            _lib.check(rc, "gms_adam_step")
        # the kernel writes through raw pointers: tell autograd the parameters changed in place (version counters feed
        # the saved-tensor modification check and the models' fused-getter caches)
        if updated:
            torch.autograd.graph.increment_version(updated)
        return loss
