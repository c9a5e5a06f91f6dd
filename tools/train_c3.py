#!/usr/bin/env python
"""BASELINE config 3 on the GPU box: the reference's training loop (train.py:39-157, restated in games_hip/train.py; the
reference's own train.training() runs on the drop-in where the reference exists: tests/test_reference_train_cpu.py) for a gs_mesh
model of the hotdog size -- 7 000 iterations, random camera order over the view stack, active SH degree raised every 1000 iterations
(train.py:86-87), fused K0 + HIP rasterizer + fused L1+SSIM loss + FusedAdam -- against 8 orbit targets rendered by a
"trained" teacher of the same mesh.  The student starts from the reference's initialisation (opacity 0.1, _scale 1, zero
higher SH, near-black colours: games/mesh_splatting/scene/gaussian_mesh_model.py:53-84).

Random camera order makes consecutive frames differ in instance count / unit count / deepest tile, so the run exercises the
decaying capacity and unit hints and (rarely) the overflow re-run of the forward tail.

    python tools/train_c3.py [--iterations 7000] [--workload c2_hotdog_like] [--views 8] > profiles/r04_train_c3_7000iters.json.log
Prints one JSON line: it/s over the whole loop (synchronised wall clock, evaluation time excluded), PSNR per view before / after,
the PSNR / loss trajectory every 500 iterations."""
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
from games_hip.model import HipGaussianMeshModel
from games_hip.render import PipelineParams, render
from games_hip.train import OptimizationParamsMesh, training


def psnr(a, b):
    mse = float(((a - b) ** 2).mean())
    return 10.0 * math.log10(1.0 / mse) if mse > 0 else float("inf")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=7000)
    ap.add_argument("--workload", default="c2_hotdog_like")
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--torch-adam", action="store_true")
    ap.add_argument("--vertices-lr", type=float, default=0.00016, help="reference default 0.0 (arguments_games/__init__.py:20)")
    ap.add_argument("--psnr-every", type=int, default=500)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    # safe_state (utils/general_utils.py:203-213)
    random.seed(0)
    torch.manual_seed(0)
    teacher_scene = syn.mesh_scene(args.workload, state="trained")
    size = teacher_scene.meta["image"]
    cams = [syn.orbit_camera(k, n_views=args.views, width=size, height=size).to(dev) for k in range(args.views)]
    bg = torch.ones(3, device=dev)
    pipe = PipelineParams()
    teacher = HipGaussianMeshModel.from_scene(teacher_scene, dev)
    with torch.no_grad():
        for c in cams:
            c.original_image = render(c, teacher, pipe, bg)["render"].clone()      # scene/cameras.py: `original_image`
    del teacher
    student = HipGaussianMeshModel.from_scene(syn.mesh_scene(args.workload, state="init"), dev)
    with torch.no_grad():
        student._features_dc.mul_(0.0)                      # near-black start, as the reader's random/255 colours
    student.active_sh_degree = 0
    opt = OptimizationParamsMesh(iterations=args.iterations, vertices_lr=args.vertices_lr)
    student.training_setup(vertices_lr=opt.vertices_lr, alpha_lr=opt.alpha_lr, feature_lr=opt.feature_lr, opacity_lr=opt.opacity_lr,
                           scaling_lr=opt.scaling_lr, fused=not args.torch_adam)

    def evaluate():
        with torch.no_grad():
            return [psnr(render(c, student, pipe, bg)["render"], c.original_image) for c in cams]

    before = evaluate()
    trajectory, n_seen = [], set()
    eval_s = [0.0]

    def report(iteration, loss_value):                      # the place of training_report (train.py:120-122); its time is not counted
        torch.cuda.synchronize()
        t = time.perf_counter()
        ps = evaluate()
        n_seen.add(last_stats()["num_rendered"])
        trajectory.append({"iteration": iteration, "loss": round(loss_value, 5), "psnr_mean": round(sum(ps) / len(ps), 2),
                           "active_sh_degree": student.active_sh_degree})
        torch.cuda.synchronize()
        eval_s[0] += time.perf_counter() - t

    at = sorted(set([1] + list(range(args.psnr_every, args.iterations + 1, args.psnr_every)) + [args.iterations]))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    import diff_gaussian_rasterization as dgr
    # the count of every training frame is read at the start of its backward (opt-in; the loop redoes a step whose
    # frame overflowed); GMS_DEFER_COUNTS=0 keeps the blocking read-back inside the forward
    deferred = os.environ.get("GMS_DEFER_COUNTS", "1") != "0" and not dgr.deterministic() and dgr._C is not None
    if deferred:
        dgr.set_deferred_counts(True)
    training(student, cams, opt, pipe, bg, report=report, report_iterations=at)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0 - eval_s[0]
    after = evaluate()
    import diff_gaussian_rasterization as dgr
    print(json.dumps({
        "what": "BASELINE config 3: gs_mesh full train loop (games_hip/train.py = train.py:39-157) on the HIP stack: fused K0, "
                "rasterizer fwd+bwd, fused L1+SSIM, " + ("torch Adam" if args.torch_adam else "FusedAdam"),
        "workload": f"{args.workload}: {student.get_xyz.shape[0]} Gaussians, {size}x{size}, {args.views} orbit targets from a teacher",
        "iterations": args.iterations, "seconds": round(el, 3), "iters_per_s": round(args.iterations / el, 1),
        "ms_per_iter": round(1000 * el / args.iterations, 4), "evaluation_seconds_excluded": round(eval_s[0], 3),
        "active_sh_degree_end": student.active_sh_degree, "deterministic_mode": bool(dgr.deterministic()), "count_readback": "deferred" if deferred else "blocking",
        "psnr_before": [round(p, 2) for p in before], "psnr_after": [round(p, 2) for p in after],
        "psnr_mean_before": round(sum(before) / len(before), 2), "psnr_mean_after": round(sum(after) / len(after), 2),
        "trajectory": trajectory, "distinct_instance_counts_sampled": len(n_seen),
        "finite": bool(all(torch.isfinite(p).all() for p in student.parameters()))}), flush=True)


if __name__ == "__main__":
    main()
