"""CPU oracle for the photometric loss -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

Restates utils/loss_utils.py in plain torch so it runs in float64 and differentiates with autograd:
  l1_loss  :17-18   mean |x - y|
  gaussian :23-25   11 taps exp(-(i-5)^2 / (2*1.5^2)), float32, normalised
  window   :27-31   outer product, one per channel (conv2d groups=channel)
  _ssim    :42-63   five zero-padded convolutions, C1=0.01^2, C2=0.03^2, mean over everything
  train.py :106-107 loss = (1-l)*l1 + l*(1-ssim)
Pinned: tests/golden/loss.npz holds outputs of the reference's own functions (tests/golden/make_golden.py)."""
from math import exp

import torch
import torch.nn.functional as F


def window_1d(dtype=torch.float32):
    g = torch.tensor([exp(-(x - 5) ** 2 / float(2 * 1.5 ** 2)) for x in range(11)], dtype=torch.float32)
    return (g / g.sum()).to(dtype)


def ssim(img1, img2):
    lead = img1.shape[:-2]
    x = img1.reshape((1, -1) + tuple(img1.shape[-2:]))
    y = img2.reshape((1, -1) + tuple(img2.shape[-2:]))
    ch = x.shape[1]
    w1 = window_1d().unsqueeze(1)
    w2 = w1.mm(w1.t()).float().to(x.dtype)                     # float32 outer product, as upstream
    win = w2.expand(ch, 1, 11, 11).contiguous().to(x.device)
    conv = lambda t: F.conv2d(t, win, padding=5, groups=ch)
    mu1, mu2 = conv(x), conv(y)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = conv(x * x) - mu1_sq
    s2 = conv(y * y) - mu2_sq
    s12 = conv(x * y) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1_mu2 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    del lead
    return m.mean()


def l1_loss(a, b):
    return torch.abs(a - b).mean()


def l1_ssim_loss(img, gt, lambda_dssim=0.2):
    return (1.0 - lambda_dssim) * l1_loss(img, gt) + lambda_dssim * (1.0 - ssim(img, gt))
