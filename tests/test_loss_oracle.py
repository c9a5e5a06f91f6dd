"""CPU: the torch restatement of the photometric loss (oracle/loss_oracle.py) against fixtures produced by executing
the reference's utils/loss_utils.py (tests/golden/make_golden.py -> loss.npz)."""
import os

import numpy as np
import pytest
import torch

from oracle import loss_oracle


@pytest.fixture(scope="module")
def gold(golden_dir):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(golden_dir, "loss.npz")).items()}


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_matches_reference_fixture(gold, tag):
    img = gold[f"img_{tag}"].clone().requires_grad_(True)
    gt = gold[f"gt_{tag}"]
    l1, ss = loss_oracle.l1_loss(img, gt), loss_oracle.ssim(img, gt)
    loss = loss_oracle.l1_ssim_loss(img, gt, 0.2)
    assert abs(float(l1) - float(gold[f"l1_{tag}"])) < 1e-7
    assert abs(float(ss) - float(gold[f"ssim_{tag}"])) < 2e-6
    assert abs(float(loss) - float(gold[f"loss_{tag}"])) < 1e-6
    loss.backward()
    ref = gold[f"d_img_{tag}"]
    assert float((img.grad - ref).abs().max()) < 1e-6 * float(ref.abs().max()) + 1e-9


def test_float64_oracle_close_to_float32_reference(gold):
    img = gold["img_a"].double().requires_grad_(True)
    loss = loss_oracle.l1_ssim_loss(img, gold["gt_a"].double(), 0.2)
    assert abs(float(loss) - float(gold["loss_a"])) < 1e-6
    loss.backward()
    ref = gold["d_img_a"].double()
    assert float((img.grad - ref).abs().max()) < 2e-4 * float(ref.abs().max())


def test_identical_images_have_unit_ssim_and_zero_l1():
    x = torch.rand(3, 48, 48)
    assert abs(float(loss_oracle.ssim(x, x)) - 1.0) < 1e-6
    assert float(loss_oracle.l1_loss(x, x)) == 0.0
