"""Fused photometric loss of the reference's training loop on MI355X (csrc/loss.hip).

Mirrors utils/loss_utils.py: `l1_loss` (:17-18), `ssim` (:33-40, window 11, sigma 1.5, size_average=True) and the
combination of train.py:106-107 as ONE forward and ONE backward kernel:

    loss = l1_ssim_loss(image, gt_image, opt.lambda_dssim)          # == (1-l)*l1_loss + l*(1-ssim)
    Ll1  = l1_ssim_loss.last_l1                                      # the value train.py logs (device scalar)

GPU tensors only; there is no CPU path in the product (the torch restatement lives in oracle/loss_oracle.py)."""
import ctypes as C

import torch

import diff_gaussian_rasterization as _dgr
from diff_gaussian_rasterization import _lib


def _weighted(img, gt, w_l1, w_ssim, bias):
    """-> (value [0-dim, differentiable], l1, ssim [device scalars, reported only])"""
    if _dgr._C is not None:          # autograd node in C++ (csrc/torch_binding.cpp), same C ABI underneath
        if img.shape != gt.shape:
            raise ValueError(f"image shapes differ: {tuple(img.shape)} vs {tuple(gt.shape)}")
        # (the node hands the value out as a 0-dim tensor of its own: indexing a [3] result would cost every backward a zeros + copy)
        value, stats = _dgr._C.l1_ssim(img, gt, float(w_l1), float(w_ssim), float(bias))
        return value, stats[0], stats[1]
    out = _WeightedL1Ssim.apply(img, gt, w_l1, w_ssim, bias)
    return out[0], out[1].detach(), out[2].detach()


_seed_cache = {}


def backward_seed(loss: torch.Tensor) -> torch.Tensor:
    """A cached `ones_like(loss)` for `loss.backward(backward_seed(loss))`: autograd otherwise launches a one-element fill per step."""
    key = (loss.device, loss.dtype, tuple(loss.shape))
    one = _seed_cache.get(key)
    if one is None:
        one = _seed_cache[key] = torch.ones(loss.shape, dtype=loss.dtype, device=loss.device)
    return one


def _planes(img1, img2):
    if img1.shape != img2.shape:
        raise ValueError(f"image shapes differ: {tuple(img1.shape)} vs {tuple(img2.shape)}")
    if img1.dim() < 2:
        raise ValueError("images must be [..., H, W]")
    h, w = int(img1.shape[-2]), int(img1.shape[-1])
    planes = 1
    for d in img1.shape[:-2]:
        planes *= int(d)
    return planes, h, w


class _WeightedL1Ssim(torch.autograd.Function):
    """value = w_l1 * mean|img-gt| + w_ssim * mean(ssim_map) + bias; returns float32[3] = (value, l1, ssim)."""

    @staticmethod
    def forward(ctx, img, gt, w_l1, w_ssim, bias):
        _lib.require_gpu(img, gt)
        lib = _lib.load()
        x = img.detach().to(torch.float32).contiguous()
        y = gt.detach().to(torch.float32).contiguous()
        planes, h, w = _planes(x, y)
        need_grad = img.requires_grad
        with _lib.on_device(x.device):
            out = torch.empty(3, dtype=torch.float32, device=x.device)
            partials = torch.empty(lib.gms_l1_ssim_partials(planes, h, w), dtype=torch.float32, device=x.device)
            dmaps = torch.empty((3,) + tuple(x.shape), dtype=torch.float32, device=x.device) if need_grad else None
            args = _lib.LossArgs(planes=planes, height=h, width=w, img=_lib.ptr(x), gt=_lib.ptr(y), w_l1=w_l1, w_ssim=w_ssim,
                                 bias=bias)
            rc = lib.gms_l1_ssim_forward(C.byref(args), _lib.ptr(dmaps), _lib.ptr(partials), _lib.ptr(out),
                                         C.c_void_p(_lib.stream_ptr(x.device)))
        _lib.check(rc, "gms_l1_ssim_forward")
        ctx.save_for_backward(x, y, dmaps if dmaps is not None else torch.empty(0, device=x.device))
        ctx.meta = (planes, h, w, w_l1, w_ssim, bias, img.dtype)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, y, dmaps = ctx.saved_tensors
        planes, h, w, w_l1, w_ssim, bias, dtype = ctx.meta
        if dmaps.numel() == 0:
            raise RuntimeError("l1_ssim backward called but the forward ran without requires_grad")
        lib = _lib.load()
        with _lib.on_device(x.device):
            # only element 0 (the value) is differentiable; l1 / ssim by-products are reported, not trained on
            g = grad_out[0:1].to(torch.float32).contiguous()
            d_img = torch.empty_like(x)
            args = _lib.LossArgs(planes=planes, height=h, width=w, img=_lib.ptr(x), gt=_lib.ptr(y), w_l1=w_l1, w_ssim=w_ssim,
                                 bias=bias)
            rc = lib.gms_l1_ssim_backward(C.byref(args), _lib.ptr(dmaps), _lib.ptr(g), _lib.ptr(d_img),
                                          C.c_void_p(_lib.stream_ptr(x.device)))
        _lib.check(rc, "gms_l1_ssim_backward")
        return d_img.to(dtype), None, None, None, None


class _L1SsimLoss:
    """Callable with the last by-products attached (device scalars; reading them does not add kernels)."""
    last_l1 = None
    last_ssim = None

    def __call__(self, image, gt_image, lambda_dssim: float = 0.2):
        lam = float(lambda_dssim)
        value, self.last_l1, self.last_ssim = _weighted(image, gt_image, 1.0 - lam, -lam, lam)
        return value


l1_ssim_loss = _L1SsimLoss()


def ssim(img1, img2, window_size=11, size_average=True):
    """utils/loss_utils.py:33 -- differentiable w.r.t. img1 (the reference only ever differentiates the render)."""
    if window_size != 11:
        raise NotImplementedError("the HIP kernel is specialised for the reference's window_size=11")
    if not size_average:
        raise NotImplementedError("size_average=False is not used by the reference's training or metrics")
    return _weighted(img1, img2, 0.0, 1.0, 0.0)[0]


def l1_loss(network_output, gt):
    """utils/loss_utils.py:17."""
    return _weighted(network_output, gt, 1.0, 0.0, 0.0)[0]
