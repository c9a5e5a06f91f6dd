"""The training loop of train.py:39-157, for the mesh-bound models, on the HIP stack.

Host-side mirror of the reference's `training()` (BASELINE config 3: "gs_mesh hotdog full train loop 7k iters").  On a machine
that holds the reference, its own `train.py` runs unchanged on the drop-in packages (tests/test_reference_train_cpu.py drives
`train.training()` itself, and shows that this restatement walks the same parameter trajectory); this module exists because the
GPU box has no reference tree.  Per iteration, in the reference's order:

    gaussians.update_learning_rate(iteration)                     train.py:83   (a no-op for gs_mesh: gaussian_mesh_model.py:185-187)
    every 1000 iterations: gaussians.oneupSHdegree()              train.py:86-87
    viewpoint_cam = stack.pop(randint(0, len(stack) - 1))         train.py:90-92  (refilled from the train cameras when empty)
    bg = rand(3) if opt.random_background else background         train.py:98
    render_pkg = render(viewpoint_cam, gaussians, pipe, bg)       train.py:100
    loss = (1 - l) * l1_loss + l * (1 - ssim)                     train.py:105-107
    loss.backward()                                               train.py:108
    [densification: gs / gs_flat only]                            train.py:129-145  (not executed for mesh models)
    optimizer.step(); optimizer.zero_grad(set_to_none=True)       train.py:147-150  (skipped on the last iteration, as there)
    gaussians.update_alpha(); gaussians.prepare_scaling_rot()     train.py:154-157

What is left out, and why: the network GUI (train.py:64-80), tensorboard / checkpoint files and `scene.save` (I/O; the `report`
callback receives what `training_report` receives), `os.makedirs(.../xyz)` every iteration."""
from __future__ import annotations

from dataclasses import dataclass
from random import randint
from typing import Callable, List, Optional, Sequence

import torch


@dataclass
class OptimizationParamsMesh:
    """arguments_games/__init__.py:17-29 (the reference's defaults: `vertices_lr` is 0.0 there, 0.00016 in its comment)."""
    iterations: int = 30_000
    vertices_lr: float = 0.0
    alpha_lr: float = 0.001
    feature_lr: float = 0.0025
    opacity_lr: float = 0.05
    scaling_lr: float = 0.005
    rotation_lr: float = 0.001
    random_background: bool = False
    use_mesh: bool = True
    lambda_dssim: float = 0.2


def training(gaussians, train_cameras: Sequence, opt: OptimizationParamsMesh, pipe, background: torch.Tensor, *,
             render: Optional[Callable] = None, loss_fn: Optional[Callable] = None, first_iter: int = 0,
             report: Optional[Callable] = None, report_iterations: Sequence[int] = ()) -> List[float]:
    """Runs iterations first_iter+1 .. opt.iterations.  `train_cameras`: objects with the camera attributes `render()` reads and
    `original_image` [3,H,W] (scene/cameras.py:17-58).  `gaussians`: a model with `training_setup()` already called (train.py:46).
    `render` / `loss_fn(image, gt, lambda_dssim) -> scalar` default to the HIP path (games_hip.render.render, the fused L1+SSIM of
    csrc/loss.hip).  `report(iteration, loss)` is called inside `torch.no_grad()` at `report_iterations` (the place of
    `training_report`, train.py:120-122).  Returns the loss values it synchronised on (one per report).  Uses `random.randint` and
    `torch.rand` exactly where the reference does: seed them as `safe_state` does (utils/general_utils.py:203-213) to reproduce its
    camera order."""
    if render is None:
        from .render import render as render_fn
    else:
        render_fn = render
    if loss_fn is None:
        from .loss import l1_ssim_loss as loss_fn
    from .loss import backward_seed as _seed
    # K0 deferred into the rasterizer's preprocess thread while the restated render() is in use (games_hip.model.HipMeshMixin.hip_defer_k0;
    # GMS_TRAIN_FUSED=0 keeps the eager launch): update_alpha() / prepare_scaling_rot() below then only mark the model stale
    import os as _os
    if hasattr(gaussians, "hip_defer_k0") and render is None and _os.environ.get("GMS_TRAIN_FUSED", "1") != "0":
        gaussians.hip_defer_k0 = True
    viewpoint_stack = None
    reported: List[float] = []
    report_at = set(int(i) for i in report_iterations)
    first_iter += 1
    for iteration in range(first_iter, opt.iterations + 1):
        if hasattr(gaussians, "update_learning_rate"):
            gaussians.update_learning_rate(iteration)
        # Every 1000 its we increase the levels of SH up to a maximum degree
        if iteration % 1000 == 0:
            gaussians.oneupSHdegree()
        # Pick a random Camera
        if not viewpoint_stack:
            viewpoint_stack = list(train_cameras)
        viewpoint_cam = viewpoint_stack.pop(randint(0, len(viewpoint_stack) - 1))
        bg = torch.rand((3), device=background.device) if opt.random_background else background
        render_pkg = render_fn(viewpoint_cam, gaussians, pipe, bg)
        image = render_pkg["render"]
        gt_image = viewpoint_cam.original_image.to(image.device)
        loss = loss_fn(image, gt_image, opt.lambda_dssim)
        try:
            loss.backward(_seed(loss))          # (train.py:108; the seed is a cached ones_like: no fill launch per iteration)
        except RuntimeError as e:
            # deferred read-back of the instance count (diff_gaussian_rasterization.set_deferred_counts): the frame outgrew its buffers and
            # that could only be seen now.  Redo the step with the blocking form (the capacity hint has been raised meanwhile).
            import diff_gaussian_rasterization as _dgr
            if _dgr.DEFERRED_OVERFLOW not in str(e):
                raise
            for group in gaussians.optimizer.param_groups:
                for p_ in group["params"]:
                    p_.grad = None
            _dgr.set_deferred_counts(False)
            try:
                if hasattr(gaussians, "update_alpha"):
                    gaussians.update_alpha()
                if hasattr(gaussians, "prepare_scaling_rot"):
                    gaussians.prepare_scaling_rot()
                render_pkg = render_fn(viewpoint_cam, gaussians, pipe, bg)
                image = render_pkg["render"]
                loss = loss_fn(image, gt_image, opt.lambda_dssim)
                loss.backward(_seed(loss))
            finally:
                _dgr.set_deferred_counts(True)
        with torch.no_grad():
            if iteration in report_at:
                reported.append(float(loss.detach()))
                if report is not None:
                    report(iteration, reported[-1])
            # Optimizer step
            if iteration < opt.iterations:
                gaussians.optimizer.step()
                gaussians.optimizer.zero_grad(set_to_none=True)
        if hasattr(gaussians, "update_alpha"):
            gaussians.update_alpha()
        if hasattr(gaussians, "prepare_scaling_rot"):
            gaussians.prepare_scaling_rot()
    return reported
