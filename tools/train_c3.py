#!/usr/bin/env python
"""BASELINE config 3 on the GPU box: the reference's training loop (train.py:39-157) for a gs_mesh model of the hotdog
size -- 7 000 iterations, random camera order over the view stack, active SH degree raised every 1000 iterations
(train.py:86-87), fused K0 + HIP rasterizer + fused L1+SSIM loss + FusedAdam -- against 8 orbit targets rendered by a
"trained" teacher of the same mesh.  The student starts from the reference's initialisation (opacity 0.1, _scale 1, zero
higher SH, near-black colours: games/mesh_splatting/scene/gaussian_mesh_model.py:53-84).

Random camera order makes consecutive frames differ in instance count / unit count / deepest tile, so the run exercises the
decaying capacity and unit hints and (rarely) the overflow re-run of the forward tail.

    python tools/train_c3.py [--iterations 7000] [--workload c2_hotdog_like] [--views 8] > profiles/r02_train_c3.json.log
Prints one JSON line: it/s over the whole loop (synchronised wall clock), PSNR per view before / after, loss curve samples."""
import argparse
import json
import math
import os
import random
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (ROOT, os.path.join(ROOT, "gaussian-mesh-splatting_amd")):
    sys.path.insert(0, p)
import torch

from diff_gaussian_rasterization import last_stats
from games_hip import synthetic as syn
from games_hip.loss import l1_ssim_loss
from games_hip.model import HipGaussianMeshModel
from games_hip.render import PipelineParams, render


def psnr(a, b):
    mse = float(((a - b) ** 2).mean())
    return 10.0 * math.log10(1.0 / mse) if mse > 0 else float("inf")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=7000)
    ap.add_argument("--workload", default="c2_hotdog_like")
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--torch-adam", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    random.seed(0)
    teacher_scene = syn.mesh_scene(args.workload, state="trained")
    size = teacher_scene.meta["image"]
    cams = [syn.orbit_camera(k, n_views=args.views, width=size, height=size).to(dev) for k in range(args.views)]
    bg = torch.ones(3, device=dev)
    pipe = PipelineParams()
    teacher = HipGaussianMeshModel.from_scene(teacher_scene, dev)
    with torch.no_grad():
        targets = [render(c, teacher, pipe, bg)["render"].clone() for c in cams]
    del teacher
    student = HipGaussianMeshModel.from_scene(syn.mesh_scene(args.workload, state="init"), dev)
    with torch.no_grad():
        student._features_dc.mul_(0.0)                      # near-black start, as the reader's random/255 colours
    student.active_sh_degree = 0
    # OptimizationParamsMesh (arguments_games/__init__.py): the reference's learning rates for gs_mesh
    student.training_setup(vertices_lr=0.00016, alpha_lr=0.001, feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005,
                           fused=not args.torch_adam)

    def evaluate():
        with torch.no_grad():
            student.update_alpha(); student.prepare_scaling_rot()
            return [psnr(render(c, student, pipe, bg)["render"], t) for c, t in zip(cams, targets)]

    before = evaluate()
    stack, curve, n_seen = [], [], set()
    ema = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(1, args.iterations + 1):
        if it % 1000 == 0:
            student.oneupSHdegree()
        if not stack:
            stack = list(range(len(cams)))
        k = stack.pop(random.randint(0, len(stack) - 1))   # train.py:88-91
        out = render(cams[k], student, pipe, bg)
        loss = l1_ssim_loss(out["render"], targets[k], 0.2)
        loss.backward()
        student.optimizer.step()
        student.optimizer.zero_grad(set_to_none=True)
        student.update_alpha()                              # train.py:154-157
        student.prepare_scaling_rot()
        if it % 500 == 0:
            lv = float(loss.detach())                                # (one sync per 500 iterations, as the reference's progress bar)
            ema = lv if ema is None else 0.6 * ema + 0.4 * lv
            curve.append([it, round(lv, 5)])
            n_seen.add(last_stats()["num_rendered"])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    after = evaluate()
    print(json.dumps({
        "what": "BASELINE config 3: gs_mesh full train loop on the HIP stack (fused K0, rasterizer fwd+bwd, fused L1+SSIM, "
                + ("torch Adam" if args.torch_adam else "FusedAdam") + ")",
        "workload": f"{args.workload}: {student.get_xyz.shape[0]} Gaussians, {size}x{size}, {args.views} orbit targets from a teacher",
        "iterations": args.iterations, "seconds": round(el, 3), "iters_per_s": round(args.iterations / el, 1),
        "ms_per_iter": round(1000 * el / args.iterations, 4), "active_sh_degree_end": student.active_sh_degree,
        "psnr_before": [round(p, 2) for p in before], "psnr_after": [round(p, 2) for p in after],
        "psnr_mean_before": round(sum(before) / len(before), 2), "psnr_mean_after": round(sum(after) / len(after), 2),
        "loss_curve": curve, "distinct_instance_counts_sampled": len(n_seen),
        "finite": bool(all(torch.isfinite(p).all() for p in student.parameters()))}), flush=True)


if __name__ == "__main__":
    main()
