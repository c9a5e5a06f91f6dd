#!/usr/bin/env python
"""bench.py -- fwd+bwd rasterizations/s of the GaMeS render path on MI355X.

One "step" = one training-style pass of the hot path over one camera view of synthetic
mesh-bound Gaussians (BASELINE.json configs[1]/[2] sizes, SURVEY.md 8(d) "C2"):
    K0 forward (mesh -> Gaussians)  ->  property getters  ->  render() forward (K1..K6)
    -> loss.backward() with dL/dcolor = (image-0.5)/(3HW)  (K7..K9, K0 backward)
    -> [N > 1] RCCL all-reduce of the parameter gradients  -> grads dropped.
N GPUs = N independent views (one per rank), weak scaling, value = views/s over the whole job.

Prints ONE JSON line (rank 0).  Extra objects: `roofline` (dominant kernel, HIP-event timed on the
launch stream), `kernels` (every kernel), `cpu_baseline` (the C oracle port on the host cores).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "gaussian-mesh-splatting_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def algorithmic_bytes(P, N, F, W, H):
    """Algorithmic HBM bytes per launch, SURVEY.md 8(d) per-unit figures (SH degree 3, scale+rotation inputs)."""
    HW = W * H
    return {
        "preprocess_fwd": 308 * P,                 # read 236 + write 72 per Gaussian
        "tile_scan": None,      # one block, 8 T bytes: a latency kernel; priced with the `binning` stage
        "emit_instances": 12 * N,                  # binning lower bound 28*N = emit 12 + sort 16
        "tile_sort": 16 * N,
        "blend_fwd": None,      # (segments 1.. of the forward walk: priced with the `forward_compositing` stage, see STAGES)
        "blend_bwd": 84 * N + 24 * HW,             # 44 + one reduced 40-B gradient record per instance
        "preprocess_bwd": 569 * P,                 # read 321 + write 248 per Gaussian
        "mesh_fwd": 36 * F + 56 * P,               # tri 36/face; alpha 12 + scale 4 in, 40 out per splat
        "mesh_bwd_splat": 56 * P,
        "mesh_bwd_face": 40 * P + 36 * F,
        # forward compositing = blend_head (first segments + products) + blend_fwd (later segments) + blend_finalize: three
        # launches of ONE stage; its algorithmic bytes (44 N + 24 HW) are priced against the SUM of the three durations
        "blend_head": None,
        "blend_finalize": None,
        "micro_filter": 52 * N,                    # micro-tile mode: key 8 + gathered record 40 per instance in, ~one id out
        "l1_ssim_fwd": 20 * 3 * HW,   # --loss l1_ssim only: read image + gt, write three derivative maps
        "l1_ssim_bwd": 24 * 3 * HW,   # read image + gt + three maps, write dL/dimage
        "adam": 28 * (3 * F // 2 + 6 + 52 * P),   # --optimizer fused_adam: 16 B read + 12 B written per parameter element
    }


def cpu_baseline(workload, state, max_seconds=12.0):
    """The oracle (C restatement of the rasterizer + torch restatement of K0) timed on the host cores."""
    from games_hip import synthetic as syn
    from oracle import gs_oracle, mesh_oracle
    # torch-CPU elementwise/gather ops of the K0 restatement stop scaling (and then collapse) beyond a few threads;
    # the C rasterizer oracle (OpenMP) is given the thread count that is fastest on this host (calibrated below:
    # on a 256-thread box 16-64 threads beat all 256)
    ncpu = os.cpu_count() or 1
    k0_threads = min(16, ncpu)
    torch.set_num_threads(k0_threads)
    sc = syn.mesh_scene(workload, state=state)
    cam = syn.orbit_camera(0, width=sc.meta["image"], height=sc.meta["image"])
    kw = dict(image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
              bg=torch.ones(3), viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=3,
              campos=cam.camera_center)
    with torch.no_grad():
        _, _, xyz, scaling, rot = mesh_oracle.mesh_to_gaussians(sc.vertices, sc.faces, sc._alpha, sc._scale)
        cal = mesh_oracle.activated(xyz, scaling, rot, sc._opacity, sc._features_dc, sc._features_rest)
    best = (float("inf"), k0_threads)
    for nt in sorted({min(n, ncpu) for n in (16, 32, 64, 128)}):
        for rep in range(2):                        # first call at a new thread count pays thread start-up
            t0 = time.time()
            o = gs_oracle.rasterize(means3D=cal[0], opacities=cal[3], shs=cal[4], scales=cal[1], rotations=cal[2], nthreads=nt, **kw)
            gs_oracle.backward(o, syn.upstream_grad(torch.from_numpy(o.color)))
            dt = time.time() - t0
        if dt < best[0]:
            best = (dt, nt)
    oracle_threads = best[1]
    times, pieces = [], {}
    t_start = time.time()
    for it in range(40):
        t0 = time.time()
        v = sc.vertices.clone().requires_grad_(True)
        a = sc._alpha.clone().requires_grad_(True)
        s = sc._scale.clone().requires_grad_(True)
        _, _, xyz, scaling, rot = mesh_oracle.mesh_to_gaussians(v, sc.faces, a, s)
        xyz_a, s_a, r_a, op_a, shs = mesh_oracle.activated(xyz, scaling, rot, sc._opacity, sc._features_dc, sc._features_rest)
        t1 = time.time()
        o = gs_oracle.rasterize(means3D=xyz_a, opacities=op_a, shs=shs, scales=s_a, rotations=r_a, nthreads=oracle_threads, **kw)
        t2 = time.time()
        g = gs_oracle.backward(o, syn.upstream_grad(torch.from_numpy(o.color)))
        t3 = time.time()
        loss = ((xyz_a * torch.from_numpy(g["means3D"])).sum() + (s_a * torch.from_numpy(g["scales"])).sum()
                + (r_a * torch.from_numpy(g["rotations"])).sum())
        loss.backward()
        t4 = time.time()
        if it > 0 or time.time() - t_start > max_seconds / 2:       # iteration 0 pays the one-time library / page warm-up
            times.append(t4 - t0)
            pieces = {"k0_fwd_s": t1 - t0, "raster_fwd_s": t2 - t1, "raster_bwd_s": t3 - t2, "k0_bwd_s": t4 - t3}
        if time.time() - t_start > max_seconds:
            break
    # north_star: "the reference's pure-PyTorch/CPU covariance-projection path timed on the host cores": the python stages the
    # reference runs with --compute_cov3D_python / --convert_SHs_python (scene/gaussian_model.py:27-31, utils/sh_utils.py:57-112)
    # and its only pure-PyTorch projection (utils/graphics_utils.py:22-29), restated device-agnostically in games_hip.render
    from games_hip.render import covariance_python, eval_sh
    torch.set_num_threads(k0_threads)

    def med(fn, n=5):
        ts = []
        for _ in range(n):
            t0 = time.time(); fn(); ts.append(time.time() - t0)
        return sorted(ts)[n // 2]

    sa, ra = cal[1].clone().requires_grad_(True), cal[2].clone().requires_grad_(True)
    shs_v = cal[4].transpose(1, 2).clone().requires_grad_(True)                     # [P,3,16] as the reference views it
    dirs = torch.nn.functional.normalize(cal[0] - cam.camera_center, dim=1)
    ones = torch.ones(cal[0].shape[0], 1)

    def cov_path():
        covariance_python(sa, 1.0, ra).sum().backward()

    def sh_path():
        torch.clamp_min(eval_sh(3, shs_v, dirs) + 0.5, 0.0).sum().backward()

    def project_path():
        h = torch.cat([cal[0], ones], dim=1) @ cam.full_proj_transform
        return h[:, :3] / (h[:, 3:4] + 0.0000001)

    stages = {"cov3D_python_fwd_bwd_s": round(med(cov_path), 4), "sh_python_fwd_bwd_s": round(med(sh_path), 4),
              "geom_transform_points_s": round(med(project_path), 4), "threads": k0_threads}
    # the dense pure-PyTorch rasterizer at config 1 (gs_flat, 10 000 Gaussians) as the "pure-PyTorch CPU
    # raster" reference point.  The dense [pixels x Gaussians] formulation keeps ~0.9 KB per pair for autograd (58 GB at 256x256):
    # a bounded sample -- 64x64 of the 256x256 pixels -- is timed and the factor to the full image stated.
    dense = None
    try:
        from oracle import dense_torch
        fs = syn.flat_scene(10000)
        c1 = syn.orbit_camera(0, width=64, height=64)
        kw1 = dict(image_height=64, image_width=64, tanfovx=c1.tanfovx, tanfovy=c1.tanfovy, bg=torch.ones(3), viewmatrix=c1.world_view_transform,
                   projmatrix=c1.full_proj_transform, sh_degree=3, campos=c1.camera_center)
        leaves = [t.clone().requires_grad_(True) for t in (fs.means3D, fs.opacities, fs.shs, fs.scales, fs.rotations)]
        t0 = time.time()
        color, _, _ = dense_torch.rasterize_dense(leaves[0], None, leaves[1], shs=leaves[2], scales=leaves[3], rotations=leaves[4], **kw1)
        t1 = time.time()
        ((color - 0.5) ** 2).sum().backward()
        t2 = time.time()
        dense = {"fwd_s": round(t1 - t0, 3), "bwd_s": round(t2 - t1, 3), "threads": k0_threads,
                 "sample": "config 1 Gaussians (gs_flat, 10 000), 64x64 of its 256x256 pixels, float32, autograd backward; "
                           "full image = 16x the pixels (memory O(P x HW): ~58 GB)", "full_image_factor": 16}
        del color, leaves
    except Exception as e:  # noqa: BLE001
        dense = {"error": repr(e)[:200]}
    t = sorted(times)[len(times) // 2]
    return {"pytorch_cov_project_path": stages, "dense_torch_raster_c1": dense,
            "k0_leg": "torch-CPU RESTATEMENT of GaussianMeshModel.update_alpha + prepare_scaling_rot (oracle/mesh_oracle.py, bit-exact against "
                      "the reference's own classes on tests/golden/k0_*.npz); the reference tree itself is not present on this box, so "
                      "BASELINE.md section 3 #1 (the imported reference classes) cannot be timed here",
            "value": 1.0 / t, "unit": "iters/s", "cores": oracle_threads, "kind": "port",
            "sample": f"{len(times)} full fwd+bwd iteration(s) of the same workload ({workload}/{state}, "
                      f"{sc.num_gaussians} Gaussians, {cam.image_width}x{cam.image_height}); C oracle with OpenMP "
                      f"({oracle_threads} threads, fastest of 16/32/64/128) + torch-CPU K0 ({k0_threads} threads), median", "host_cpu_count": os.cpu_count(), "k0_torch_threads": k0_threads, **{k: round(v, 4) for k, v in pieces.items()}}


def kernel_source_hash():
    """sha256 (16 hex) over the kernel sources, comments stripped (tools/srchash.py): profiles/pmc_traffic.json records the
    hash of the build its counters were collected on (tools/make_pmc_traffic.py); counters from another build are NOT
    reported against this build's durations."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import srchash
    finally:
        sys.path.pop(0)
    return srchash.kernel_source_hash()


# Stages that take several kernels: priced as ONE unit (sum of the launches' durations against the stage's algorithmic bytes,
# SURVEY.md 8(d)).  Round 3 booked the forward walk's bytes on its first launch and showed `frac 0.0` for the other two.
STAGES = {
    "forward_compositing": (("blend_head", "blend_fwd", "blend_finalize"), lambda P, N, F, HW: 44 * N + 24 * HW),
    "binning": (("tile_scan", "emit_instances", "tile_sort"), lambda P, N, F, HW: 28 * N),
    "k0_plus_preprocess_fwd": (("mesh_fwd", "preprocess_fwd"), lambda P, N, F, HW: 36 * F + 56 * P + 308 * P),
}

def rccl_summary(path):
    """What this rank's RCCL INIT / GRAPH log says about the communicator (never fails the bench: best effort)."""
    env = {k: os.environ[k] for k in ("NCCL_ALGO", "NCCL_PROTO", "NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS", "RCCL_MSCCL_ENABLE", "NCCL_P2P_LEVEL",
                                      "HSA_ENABLE_IPC_MODE_LEGACY") if k in os.environ}
    out = {"env": env, "algorithm_protocol": "RCCL's tuner per collective and size (NCCL_ALGO / NCCL_PROTO unset)" if "NCCL_ALGO" not in env and "NCCL_PROTO" not in env
           else f"forced: NCCL_ALGO={env.get('NCCL_ALGO')} NCCL_PROTO={env.get('NCCL_PROTO')}", "log": path}
    try:
        if not path or not os.path.exists(path):
            out["note"] = "no RCCL debug file (NCCL_DEBUG_FILE set elsewhere, or a non-RCCL backend)"
            return out
        import re
        with open(path, errors="replace") as f:
            lines = f.read().splitlines()
        txt = "\n".join(lines)
        m = re.search(r"(RCCL|NCCL) version[^\n]*", txt)
        out["version"] = m.group(0)[:120] if m else None
        m = re.search(r"nranks (\d+)", txt)
        out["nranks"] = int(m.group(1)) if m else None
        ch = re.findall(r"Channel (\d+)/(\d+)", txt)
        out["channels"] = int(ch[0][1]) if ch else None
        out["transports"] = sorted(set(re.findall(r"via (P2P/\w+|SHM\w*|NET/\w+|direct\w*)", txt)))[:8]
        out["xgmi_mentions"] = len(re.findall(r"XGMI|xgmi", txt))
        out["rings_trees"] = [ln.split("NCCL INFO", 1)[-1].strip()[:160] for ln in lines if ("Connected all" in ln or "Trees" in ln or "Ring 0" in ln)][:6]
        out["lines"] = len(lines)
    except Exception as e:  # noqa: BLE001
        out["error"] = repr(e)[:200]
    return out


BASELINE_METRIC = "train iters/s (fwd+bwd raster) @800×800, 300k Gaussians; HBM GB/s vs roofline"
# Vector-instruction issue rate, MEASURED (tools/valu_bench.hip, profiles/r06_valu_microbenchmark.txt: 8 waves per SIMD of independent
# chains): a plain f32 wave-instruction occupies a SIMD for 1.2 ns (DPP 1.76, packed / f64 1.8, transcendental 3.4) -- the chip clocks to
# its power budget under vector load, so the 2 cycles at 2.4 GHz = 0.83 ns that rounds 2-5 priced against (78.6 T lane-op/s) is not
# reachable.  256 CUs x 4 SIMDs x 64 lanes / 1.2 ns:
VALU_NS_PLAIN = 1.2
VALU_PEAK_TLANEOPS = 256 * 4 * 64 / VALU_NS_PLAIN * 1e9 / 1e12     # 54.6 T lane-op/s for plain instructions


def valu_model():
    """profiles/valu_model.json (tools/valu_mix.py): per compositing kernel the static instruction-class mix of its walk loop and the
    mix-weighted issue cost per wave-instruction; ignored when it was made from other kernel sources than this build's."""
    path = os.path.join(ROOT, "profiles", "valu_model.json")
    if not os.path.exists(path):
        return {}, "no profiles/valu_model.json"
    with open(path) as f:
        m = json.load(f)
    if m.get("_source_hash") != kernel_source_hash():
        return {}, f"profiles/valu_model.json was made from kernel sources {m.get('_source_hash')}, this build is {kernel_source_hash()}: plain-rate figures only"
    return m.get("kernels", {}), "tools/valu_mix.py on this build's sources x tools/valu_bench.hip"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1,
                    help="ranks (one GPU each).  Without a launcher (WORLD_SIZE unset) and N > 1, bench.py starts the N ranks "
                         "itself through torch.distributed.run on 127.0.0.1")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=5,
                    help="the K-step timed region (barrier + synchronize on both sides, exactly K steps) is run this many times back to "
                         "back; `value` / `ms_per_step` are the MEDIAN region, every region is listed under `repeats` (round-5 review: "
                         "with the driver's --steps 20 one region is 8 ms -- a single sample is not a record)")
    ap.add_argument("--workload", default=None,
                    help="default: c2_hotdog_like (gs_mesh, BASELINE configs[1]/[2]) on one GPU, c4_ficus_like (gs_multi_mesh, "
                         "BASELINE configs[3]: 8 views sharded one per GPU) on several")
    ap.add_argument("--state", default="trained", choices=["trained", "init"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=20)
    ap.add_argument("--mode", default="train", choices=["train", "animate"],
                    help="train: the headline fwd+bwd step; animate: BASELINE config 5 style forward-only renders with per-frame "
                         "vertex animation and on-device re-derivation of scale/rotation (secondary line, not the headline)")
    ap.add_argument("--graph", action="store_true",
                    help="--mode animate only: replay the frame as a captured hipGraph (games_hip.animate.GraphedAnimation) instead of "
                         "enqueuing it from Python every frame")
    ap.add_argument("--loss", default="dense_grad", choices=["dense_grad", "l1_ssim"],
                    help="dense_grad: the headline step of SURVEY 8(d), dL/dcolor = (image-0.5)/(3HW); l1_ssim: the reference's "
                         "training loss (train.py:106-107) through the fused HIP L1+SSIM kernels against a synthetic target")
    ap.add_argument("--views-per-step", type=int, default=1,
                    help="views each rank renders (fwd+bwd) per step, gradients accumulated before the ONE all-reduce of the step. "
                         "The headline is 1 at every N (config 4: one view per GPU per step); on several GPUs the 4-view amortised "
                         "figure is measured too and reported under `amortised`")
    ap.add_argument("--allreduce", default="auto", choices=["auto", "ring", "direct", "direct_ag"],
                    help="how the large (SH) gradient is reduced on several GPUs: ring = one all_reduce (RCCL's choice of algorithm), "
                         "direct = two all-to-all phases over all xGMI links at once (games_hip.ddp.DirectAllReduce); auto times both "
                         "on gradient-sized buffers before the timed region and takes the faster one (both times are reported)")
    ap.add_argument("--sh-exchange", default="auto", choices=["auto", "dense", "factor", "packed"],
                    help="several ranks: how the SH gradient (54 of the 64 MB) crosses ranks.  dense: inside the gradient all-reduce; "
                         "factor: all-gather of the per-view [P,3] colour-gradient factors + local expansion (gms_sh_grad_expand) next to an "
                         "all-reduce of the remaining 6.6 MB; packed: ONE all-gather of [small gradients | factors], summed locally; "
                         "auto = a short timed run of the step with each, the fastest wins (all three times are reported)")
    ap.add_argument("--no-fused-k0", action="store_true",
                    help="keep the mesh -> Gaussian step (K0) an eager launch of its own.  Default at one view per step on a single-mesh "
                         "model: update_alpha() / prepare_scaling_rot() defer it and render() derives the Gaussians inside the rasterizer's "
                         "preprocess thread (games_hip.model.HipMeshMixin.hip_defer_k0; same calls, same image, same gradients)")
    ap.add_argument("--no-defer-counts", action="store_true",
                    help="read the frame's instance count back inside the forward (the blocking form).  Default in train mode on one rank: "
                         "diff_gaussian_rasterization.set_deferred_counts(True) -- the count is read at the start of the backward, the host runs "
                         "ahead of the GPU by the loss; an overflowed frame raises there and the step is redone (counted on the line)")
    ap.add_argument("--optimizer", default="none", choices=["none", "fused_adam", "torch_adam"],
                    help="none: gradients are dropped after the (all-reduced) backward, the headline step; fused_adam / torch_adam: "
                         "also run optimizer.step() of the reference's training_setup() (train.py:147) with lr scaled to ~0 so "
                         "the scene, and with it the work per step, stays fixed")
    return ap.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU, RCCL) through torch.distributed.run.
    On a box with fewer than N GPUs (the 1-GPU test box) the ranks share cuda:0 over gloo -- flagged in the JSON."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    if torch.cuda.device_count() < args.gpus:
        env["GMS_BENCH_SHARED_GPU"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def build_model(workload, state, device):
    """(model, size, description dict).  c4_* / multi_* workloads are gs_multi_mesh, everything else gs_mesh."""
    from games_hip import synthetic as syn
    from games_hip.model import HipGaussianMeshModel, HipGaussianMultiMeshModel
    if workload in syn.MULTI_MESH_CONFIGS:
        scenes = syn.multi_mesh_scenes(workload, state=state)
        model = HipGaussianMultiMeshModel.from_scenes(scenes, device)
        F = sum(int(s.faces.shape[0]) for s in scenes)
        P = sum(s.num_gaussians for s in scenes)
        desc = {"model": "gs_multi_mesh", "gaussians": P, "faces": F, "meshes": [[int(s.faces.shape[0]), s.meta["S"]] for s in scenes],
                "text": f"{workload}/{state}: {len(scenes)} UV-sphere meshes (faces x splats: "
                        + ", ".join(f"{int(s.faces.shape[0])}x{s.meta['S']}" for s in scenes) + f") = {P} mesh-bound Gaussians"}
        return model, scenes[0].meta["image"], desc
    scene = syn.mesh_scene(workload, state=state)
    model = HipGaussianMeshModel.from_scene(scene, device)
    F, P = int(scene.faces.shape[0]), scene.num_gaussians
    desc = {"model": "gs_mesh", "gaussians": P, "faces": F,
            "text": f"{workload}/{state}: UV-sphere mesh F={F} x {scene.meta['S']} splats = {P} mesh-bound Gaussians"}
    return model, scene.meta["image"], desc


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and os.environ.get("GMS_BENCH_FORCE_DDP") != "1":
        sys.exit(self_launch(args))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # GMS_BENCH_SHARED_GPU=1 (testing the multi-rank code path on a 1-GPU box): every rank on cuda:0, gloo backend
    shared_gpu = os.environ.get("GMS_BENCH_SHARED_GPU") == "1"
    dev_index = 0 if shared_gpu else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # GMS_BENCH_FORCE_DDP=1 on one GPU: a one-rank RCCL process group with the gradient all-reduce in the step (exercises
    # the nccl code path -- communicator creation, async collectives from autograd hooks, stream waits -- without peers)
    force_ddp = world == 1 and os.environ.get("GMS_BENCH_FORCE_DDP") == "1"
    backend = None
    rccl_log = None
    if (world > 1 and not shared_gpu) or force_ddp:
        # self-diagnosing multi-GPU run: RCCL's own INIT / GRAPH log of this rank goes to a file that rank 0 summarises into the JSON
        # line (version, ranks, channels, transports, the algorithm / protocol environment) -- the first 8-GPU run explains itself
        rccl_log = f"/tmp/gms_bench_rccl_{os.getpid()}.log"
        if os.environ.get("GMS_BENCH_RCCL_LOG", "1") != "0":
            if os.environ.get("NCCL_DEBUG", "").upper() not in ("INFO", "TRACE"):
                os.environ["NCCL_DEBUG"] = "INFO"          # (a preset WARN / VERSION level would leave the file empty)
            os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH,ENV")
            os.environ.setdefault("NCCL_DEBUG_FILE", rccl_log)
        if os.environ.get("NCCL_DEBUG_FILE") != rccl_log:
            rccl_log = None
    if force_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        backend = "nccl"
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if shared_gpu else "nccl"
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    distributed = world > 1 or force_ddp

    from diff_gaussian_rasterization import _lib, keep_buffers, last_stats
    from games_hip import synthetic as syn
    from games_hip.ddp import DirectAllReduce, OverlappedGradAllReduce, PackedGradExchange, ShFactorExchange
    from games_hip.render import PipelineParams, render

    workload = args.workload or ("c2_hotdog_like" if world == 1 else "c4_ficus_like")
    model, size, desc = build_model(workload, args.state, device)
    bg = torch.ones(3, device=device)
    pipe = PipelineParams()
    params = model.parameters()
    all_cams = [syn.orbit_camera(k, width=size, height=size).to(device) for k in range(8)]

    if args.loss == "l1_ssim":
        from games_hip.loss import l1_ssim_loss, backward_seed
        yy, xx = torch.meshgrid(torch.linspace(0, 6, size, device=device), torch.linspace(0, 5, size, device=device), indexing="ij")
        gt_image = (0.5 + 0.4 * torch.sin(2.0 * xx) * torch.cos(1.5 * yy)).expand(3, size, size).contiguous()

    if args.optimizer != "none":
        model.training_setup(vertices_lr=1e-12, alpha_lr=1e-12, feature_lr=1e-12, opacity_lr=1e-12, scaling_lr=1e-12,
                             fused=args.optimizer == "fused_adam")

    def sync():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(device)

    # ---- several ranks: which collective for the large gradient (timed on gradient-sized buffers, outside the timed region)
    sh_factor = distributed and (args.sh_exchange in ("factor", "packed") or (args.sh_exchange == "auto" and (world > 1 or force_ddp)))
    sh_mode_default = ("packed" if args.sh_exchange == "packed" else "factor") if sh_factor else "dense"
    algo, allreduce_times = "ring", {}
    if distributed:
        # (factorised SH exchange: the feature tensors take no part in the all-reduce)
        dense_params = [p for p in params if not (sh_factor and (p is model._features_dc or p is model._features_rest))]
        big = [torch.zeros_like(p) for p in dense_params if p.numel() >= (1 << 22)]
        small_n = sum(p.numel() for p in dense_params if p.numel() < (1 << 22))
        flat = torch.zeros(max(small_n, 1), device=device)

        def collectives_once(which):
            works, ds = [], []
            for b in big:
                if which in ("direct", "direct_ag") and (world > 1 or force_ddp):
                    d = DirectAllReduce(world, gather="all_gather" if which == "direct_ag" else "all_to_all")
                    d.start(b)
                    ds.append(d)
                else:
                    works.append(dist.all_reduce(b, async_op=True))
            works.append(dist.all_reduce(flat, async_op=True))
            for w in works:
                w.wait()
            for d in ds:
                d.finish()

        for which in (("ring", "direct", "direct_ag") if args.allreduce == "auto" else (args.allreduce,)):
            # a local failure is caught FIRST; then ONE agreement collective runs on every rank unconditionally (same dtype,
            # same op everywhere): t = +inf on the rank that failed, MAX over ranks -> every rank reaches the same verdict
            t_local, err = float("inf"), None
            try:
                for _ in range(3):
                    collectives_once(which)
                sync()
                t0 = time.perf_counter()
                for _ in range(10):
                    collectives_once(which)
                sync()
                t_local = (time.perf_counter() - t0) / 10
            except Exception as e:  # noqa: BLE001 - an algorithm the backend cannot run is simply not chosen
                err = f"failed: {e!r}"[:200]
            t = torch.tensor([t_local], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            if math.isfinite(float(t.item())):
                allreduce_times[which] = round(1000 * float(t.item()), 4)
            else:
                allreduce_times[which] = err or "failed on another rank"
        ok = {k: v for k, v in allreduce_times.items() if isinstance(v, float)}
        algo = min(ok, key=ok.get) if ok else "ring"
        allreduce_bytes = 4 * (sum(b.numel() for b in big) + flat.numel())
        del big, flat

    import diff_gaussian_rasterization as dgr
    deferred_overflows = [0]
    defer_counts = (not args.no_defer_counts and args.mode == "train" and not distributed and dgr._C is not None and not dgr.deterministic())
    if defer_counts:
        dgr.set_deferred_counts(True)
    from games_hip.model import HipMeshMixin, HipFlameMixin
    fused_k0 = (not args.no_fused_k0 and args.mode == "train" and isinstance(model, HipMeshMixin) and not isinstance(model, HipFlameMixin))

    def make_step(vps, reduce_grads, sh_mode=None):
        # K0 inside the preprocess thread: one view per step (several views would each redo the mesh backward), no gradient exchange
        # hooked on get_xyz
        model.hip_defer_k0 = bool(fused_k0 and vps == 1 and not (reduce_grads and distributed))
        sh_mode = sh_mode_default if sh_mode is None else sh_mode
        sh_factor = sh_mode == "factor"
        """One step = K0 forward (once: the parameters are the same for all its views) + vps x (render fwd + bwd) on this rank's
        views + [reduce_grads] ONE gradient all-reduce.  Rank r renders cameras (r*vps + v) % 8 (config 4: 8 views)."""
        cams = [all_cams[(rank * vps + v) % 8] for v in range(vps)]
        # mean over ALL views of the step (this rank's and the other ranks'): the 1/world of the gradient average is folded
        # into the upstream gradient, so the all-reduce is a plain sum and no 64 MB division pass follows it
        inv_norm = 1.0 / (3.0 * size * size * vps * world)
        neg_half_norm = torch.tensor(-0.5 * inv_norm)      # (a 0-dim CPU tensor is a scalar to the elementwise kernel: a device one makes it the strided, unvectorised form)
        packed = PackedGradExchange(params, model._features_dc, model._features_rest, world, force=force_ddp, average=False) if (reduce_grads and distributed and sh_mode == "packed") else None
        reducer = OverlappedGradAllReduce(params, world, average=False, force=force_ddp, algorithm=algo) if (reduce_grads and distributed and packed is None) else None
        exchange = ShFactorExchange(model._features_dc, model._features_rest, world, force=force_ddp, average=False) if (reducer is not None and sh_factor) else None

        host_sleep_s = 1e-6 * float(os.environ.get("GMS_BENCH_HOST_SLEEP_US", "0"))      # throttled-host experiment: busy host time per step

        def step():
            try:
                step_once()
            except RuntimeError as e:          # a deferred frame outgrew its buffers: redo the step with the blocking form
                if dgr.DEFERRED_OVERFLOW not in str(e):
                    raise
                deferred_overflows[0] += 1
                for p in params:
                    p.grad = None
                dgr.set_deferred_counts(False)
                try:
                    step_once()
                finally:
                    dgr.set_deferred_counts(True)
            if host_sleep_s > 0.0:
                t_end = time.perf_counter() + host_sleep_s
                while time.perf_counter() < t_end:
                    pass

        def step_once():
            if packed is not None:
                packed.enable()
            if exchange is not None:
                exchange.enable()
            model.update_alpha()
            model.prepare_scaling_rot()
            images = [render(c, model, pipe, bg)["render"] for c in cams]
            if exchange is not None:
                exchange.watch(model.get_xyz)     # the gather starts inside backward, right after the last rasterizer backward
            if args.loss == "l1_ssim":
                loss = l1_ssim_loss(images[0], gt_image, 0.2)
                for im in images[1:]:
                    loss = loss + l1_ssim_loss(im, gt_image, 0.2)
                loss = loss / (vps * world) if vps * world > 1 else loss
                loss.backward(backward_seed(loss))          # (as games_hip.train does: a cached ones_like instead of a fill launch per step)
            else:
                with torch.no_grad():                        # dL/dcolor = (image - 0.5) / (3HW), dense
                    grads = [torch.add(neg_half_norm, im, alpha=inv_norm) for im in images]      # one elementwise kernel each
                torch.autograd.backward(images, grads)
            if reducer is not None:
                reducer.finish()      # collectives were started from autograd hooks during backward
            if exchange is not None:
                exchange.finish(model.get_xyz, model.active_sh_degree)
                exchange.disable()
            if packed is not None:
                packed.finish(model.get_xyz, model.active_sh_degree)      # the step's ONE collective
                packed.disable()
            if args.optimizer != "none":
                model.optimizer.step()
                model.optimizer.zero_grad(set_to_none=True)
            else:
                for p in params:
                    p.grad = None
        return step, reducer

    rank_spread = {}

    def timed(step, steps, warmup, tag=None):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; MAX over ranks.  `tag`: also record
        every rank's own elapsed time and its time to reach the closing barrier (per-rank spread: a straggler shows here)."""
        for _ in range(warmup):
            step()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize(device)
        own = time.perf_counter() - t0          # this rank's K steps, before waiting for the others
        sync()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
            if tag is not None:
                mine = torch.tensor([own], device=device, dtype=torch.float64)
                every = [torch.zeros_like(mine) for _ in range(world)]
                dist.all_gather(every, mine)
                ms = [1000.0 * float(x.item()) / steps for x in every]
                rank_spread[tag] = {"ms_per_step_by_rank": [round(v, 4) for v in ms], "min": round(min(ms), 4), "max": round(max(ms), 4),
                                    "spread_frac": round((max(ms) - min(ms)) / max(ms), 4) if max(ms) > 0 else 0.0}
        return el

    if args.mode == "animate":
        from games_hip.render import _fused_frame_ok, render_animated
        from games_hip.animate import render_frame
        cam = all_cams[rank % 8]
        frame = [0]
        verts = torch.cat(list(model.vertices)) if isinstance(model.vertices, (list, tuple)) else model.vertices
        faces = model._hip_topology()[0] if isinstance(model.faces, (list, tuple)) else model.faces

        anim = None
        if args.graph:
            from games_hip.animate import GraphedAnimation
            anim = GraphedAnimation(model, cam, pipe, bg)

        with torch.no_grad():
            fused_k0 = (not isinstance(model.faces, (list, tuple))) and _fused_frame_ok(model, pipe, None)

        def animate_step():
            with torch.no_grad():
                t = 0.05 * frame[0]
                frame[0] += 1
                new_v = verts * (1.0 + 0.05 * math.sin(t))           # scripts/render_time_animated.py:68-87 style
                if anim is not None:
                    anim.render(new_v[faces], check=(frame[0] % 64 == 0))      # the frame's counts are read back every 64th frame
                elif fused_k0:
                    render_frame(new_v, faces, cam, model, pipe, bg)         # mesh -> image: K0 inside the preprocess thread
                else:
                    render_animated(None, new_v[faces], cam, model, pipe, bg)
        el = timed(animate_step, args.steps, args.warmup)
        if rank == 0:
            print(json.dumps({"metric": "renders/s (fwd only, per-frame vertex animation + fused face->Gaussian + raster)",
                              "value": round(world * args.steps / el, 2), "unit": "renders/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * el / args.steps, 4),
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                              "config": {"workload": f"{desc['text']}, animate, {size}x{size}",
                                         "k0": ("inside preprocess_fwd (GmsRasterForwardArgs.mesh): no K0 launch, no vertices[faces] gather"
                                                if (fused_k0 and anim is None) else ("inside preprocess_fwd, frame replayed as a hipGraph" if fused_k0 else "separate launch")),
                                         "graph": ({"captures": anim.captures, **anim.status()} if anim is not None else None)}}), flush=True)
        if distributed:
            dist.destroy_process_group()
        return

    vps = max(1, args.views_per_step)
    sh_exchange_ms = {}
    if distributed and args.sh_exchange == "auto":
        # like the all-reduce algorithm: measured, not assumed -- a short run of the step with each exchange, the same on every rank
        for mode in ("dense", "factor", "packed"):
            s_try, r_try = make_step(vps, True, sh_mode=mode)
            el_try = timed(s_try, 30, 10)
            sh_exchange_ms[mode] = round(1000 * el_try / 30, 4)
            if r_try is not None:
                r_try.remove()
        sh_mode_default = min(sh_exchange_ms, key=sh_exchange_ms.get)
        sh_factor = sh_mode_default != "dense"
    step, reducer = make_step(vps, True)
    # untimed pre-warm (~0.3 s of steps before the W warm-up steps): allocator pools, capacity / unit hints and the GPU's
    # clocks reach their steady state; two back-to-back runs on one box otherwise differ by 6 % (first run slower)
    if distributed:                      # the same number of steps on every rank: each step contains collectives
        for _ in range(int(os.environ.get("GMS_BENCH_PREWARM_STEPS", "300"))):
            step()
        torch.cuda.synchronize(device)
    else:
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < float(os.environ.get("GMS_BENCH_PREWARM_S", "0.3")):
            for _ in range(10):
                step()
            torch.cuda.synchronize(device)
    _lib.load().gms_wait_stats(None, None, 1)          # (reset: host time spent polling for N inside the timed regions)
    regions = [timed(step, args.steps, args.warmup, tag="headline")]
    for _ in range(max(1, args.repeats) - 1):
        regions.append(timed(step, args.steps, 0))
    import ctypes as _Cw
    _wms, _wcalls = _Cw.c_double(0.0), _Cw.c_int64(0)
    _lib.load().gms_wait_stats(_Cw.byref(_wms), _Cw.byref(_wcalls), 0)
    host_wait_us_per_step = 1000.0 * _wms.value / max(1, args.steps * len(regions))
    elapsed = sorted(regions)[len(regions) // 2]          # the median region (each one: exactly K steps, bracketed as the contract says)
    ms_per_step = 1000.0 * elapsed / args.steps
    value = world * vps * args.steps / elapsed
    keep_buffers(True)          # one untimed step whose scratch stays referenced: visible count / interactions for the JSON
    step()
    stats = last_stats()
    keep_buffers(False)

    # ---- several ranks: what the collective costs, measured in the same job
    extra = {}
    if distributed:
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)
        extra["ranks_seen"] = int(ones.item())
        extra["per_rank_step_time"] = rank_spread.get("headline")
        extra["backend"] = backend + ("/shared-gpu (test path: every rank on cuda:0)" if shared_gpu else "/RCCL over xGMI" if backend == "nccl" else "")
        if reducer is not None:
            reducer.remove()
        # (a) the same step without the gradient exchange (= N independent replicas): the per-GPU rate the weak-scaling
        #     efficiency is measured against, same workload, same job
        s_nc, _ = make_step(vps, False)
        el_nc = timed(s_nc, args.steps, 2)
        extra["no_comm"] = {"value": round(world * vps * args.steps / el_nc, 2), "ms_per_step": round(1000 * el_nc / args.steps, 4),
                            "note": "same step without the gradient all-reduce (independent replicas)"}
        extra["efficiency_vs_no_comm"] = round(el_nc / elapsed, 4)
        # (b) four views per rank per step, one all-reduce per step ("fewer, larger collectives"): the amortised figure
        if vps == 1:
            s_am, r_am = make_step(4, True)
            el_am = timed(s_am, max(1, args.steps // 2), 2)
            extra["amortised"] = {"views_per_rank_per_step": 4, "value": round(world * 4 * max(1, args.steps // 2) / el_am, 2),
                                  "ms_per_step": round(1000 * el_am / max(1, args.steps // 2), 4)}
            if r_am is not None:
                r_am.remove()
        # (c) the collectives alone on gradient-sized buffers (one large + one flat bucket, as the reducer issues them)
        extra["allreduce_ms"] = allreduce_times.get(algo)
        extra["allreduce_algorithms_ms"] = allreduce_times
        extra["allreduce_algorithm"] = algo
        extra["allreduce_bytes"] = allreduce_bytes
        extra["sh_exchange"] = {"factor": "factor: all-gather of [P+1,3] colour-gradient factors + gms_sh_grad_expand, all-reduce of the rest",
                                "packed": "packed: ONE all-gather of [small gradients | colour-gradient factors], summed and expanded locally",
                                "dense": "dense: the SH gradient travels inside the all-reduce"}[sh_mode_default]
        extra["sh_exchange_ms_per_step"] = sh_exchange_ms or None
        gather_bytes = 4 * 3 * (int(model.get_xyz.shape[0]) + 1) * vps * world if sh_factor else 0
        if sh_factor:
            extra["sh_factor_gather_bytes"] = gather_bytes
        dense_numel = sum(p.numel() for p in params if not (sh_factor and (p is model._features_dc or p is model._features_rest)))
        # per rank per step: all-reduced gradient + gathered factors (packed: everything gathered, W x the small gradients)
        extra["exchange_bytes"] = (4 * dense_numel * world + gather_bytes) if sh_mode_default == "packed" else (4 * dense_numel + gather_bytes)
        extra["exchange_candidates"] = {"sh_exchange_ms_per_step": sh_exchange_ms or None, "allreduce_algorithms_ms": allreduce_times,
                                        "chosen": {"sh_exchange": sh_mode_default, "allreduce": algo}}
        step, reducer = make_step(vps, True)           # for the profiling pass below

    # ---- per-kernel durations: HIP events on the launch stream (separate untimed pass)
    lib = _lib.load()
    lib.gms_profile_reset()
    lib.gms_profile_enable(1)
    for _ in range(args.profile_steps):
        step()
    torch.cuda.synchronize(device)
    lib.gms_profile_enable(0)
    ktimes = _lib.kernel_times()
    # what the bracketing event pair itself adds to every launch it times (round-4 review: without this the small kernels read up
    # to 32 % long and the table sums to more than the step): measured here, on this stream, subtracted below
    event_overhead_us = 0.0
    if args.profile_steps > 0:
        import ctypes as _C
        event_overhead_us = max(0.0, float(lib.gms_profile_event_overhead_us(_C.c_void_p(torch.cuda.current_stream(device).cuda_stream), 40)))

    if rank == 0:
        P, F = desc["gaussians"], desc["faces"]
        N = int(stats.get("num_rendered", 0))
        ab = algorithmic_bytes(P, N, F, size, size)
        pmc = {}
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        pmc_note = "no PMC counters committed for this workload"
        if os.path.exists(pmc_path):
            with open(pmc_path) as f:
                pmc_all = json.load(f)
            pmc = pmc_all.get(f"{workload}/{args.state}", {})
            have, want = pmc.get("_source_hash"), kernel_source_hash()
            if pmc and have != want:
                # counters of another build say nothing about this build's kernels: report durations only
                pmc_note = f"profiles/pmc_traffic.json was collected on kernel sources {have}, this build is {want}: counters withheld"
                pmc = {}
            elif pmc:
                pmc_note = f"separate rocprofv3 --pmc passes of this workload on kernel sources {have} (= this build)"
        sq = pmc.get("_sq", {})
        if ktimes.get("mesh_bwd_splat", (0, 0))[1] == 0:       # K0 backward ran as one fused launch (booked as mesh_bwd_face)
            ab["mesh_bwd_face"] += ab["mesh_bwd_splat"]
        k0_bwd_inside = (getattr(model, "hip_defer_k0", False) and ktimes.get("mesh_bwd_face", (0, 0))[1] == 0
                         and ktimes.get("preprocess_bwd", (0, 0))[1] > 0)
        if k0_bwd_inside:
            # frames rendered from the mesh (ABI 8): preprocess_bwd carries the gradients on through the face -> Gaussian parameterization
            # itself -- dL/dxyz, dL/dscale, dL/drot, dL/dopacity (44 B) are not written, per splat _alpha / _scale / _opacity come in and
            # their gradients go out (40 B), per face the indices and corners come in and the corner gradients go out (96 B)
            ab["preprocess_bwd"] = (569 - 44 + 40) * P + 96 * F
        kernels = {}
        for name, (ms, n) in ktimes.items():
            if n == 0:
                continue
            raw_us = 1000.0 * ms / n
            avg_us = max(raw_us - event_overhead_us, 0.25 * raw_us)      # (the floor only guards a launch shorter than the overhead)
            lps = n / max(args.profile_steps * vps, 1)       # launches per rendered view
            # a stage that takes several launches per view (tile_sort: presort + merge; blend_head on deep scenes) is priced on
            # the SUM of its launches: its algorithmic bytes are per view, not per launch
            if ab[name] is None:          # one launch of a multi-kernel stage: priced under `stages`, not here
                kernels[name] = {"avg_us": round(avg_us, 2), "launches_per_step": n / max(args.profile_steps, 1),
                                 "algorithmic_bytes": None, "achieved_GBps": None, "frac_of_8TBps": None, "traffic": pmc.get(name),
                                 "priced_with": next(k for k, v in STAGES.items() if name in v[0])}
            else:
                gbs = ab[name] / (avg_us * max(lps, 1.0) * 1e-6) / 1e9
                kernels[name] = {"avg_us": round(avg_us, 2), "launches_per_step": n / max(args.profile_steps, 1),
                                 "algorithmic_bytes": ab[name], "achieved_GBps": round(gbs, 1),
                                 "frac_of_8TBps": round(gbs / 8000.0, 4), "traffic": pmc.get(name)}
            if name in sq and sq[name].get("SQ_INSTS_VALU"):
                lane_ops = sq[name]["SQ_INSTS_VALU"] * 64.0
                kernels[name]["valu_wave_insts"] = sq[name]["SQ_INSTS_VALU"]
                # wave-instructions x the measured 1.2 ns of a PLAIN instruction / 1 024 SIMDs: a lower bound of the issue time
                kernels[name]["valu_issue_frac"] = round(lane_ops / (avg_us * 1e-6) / 1e12 / VALU_PEAK_TLANEOPS, 4)
        stages = {}
        for sname, (members, fn) in STAGES.items():
            have_k = [m for m in members if m in kernels]
            if not have_k:
                continue
            us = sum(kernels[m]["avg_us"] * kernels[m]["launches_per_step"] for m in have_k) / max(vps, 1)
            sb = fn(P, N, F, size * size)
            # (PMC traffic is per launch -- for tile_sort the mean of its launches: a stage's traffic counts every launch of the view)
            tr = [None if kernels[m]["traffic"] is None else int(kernels[m]["traffic"] * kernels[m]["launches_per_step"] / max(vps, 1)) for m in have_k]
            stages[sname] = {"kernels": have_k, "sum_us_per_view": round(us, 2), "algorithmic_bytes": sb,
                             "achieved_GBps": round(sb / (us * 1e-6) / 1e9, 1) if us > 0 else None,
                             "frac_of_8TBps": round(sb / (us * 1e-6) / 8e12, 4) if us > 0 else None,
                             "traffic": sum(tr) if all(t is not None for t in tr) else None}
        if not kernels:     # --profile-steps 0: no per-kernel timing requested
            kernels = {"(not profiled)": {"avg_us": 0.0, "launches_per_step": 0, "algorithmic_bytes": 0, "achieved_GBps": 0.0,
                                          "frac_of_8TBps": 0.0, "traffic": None}}
        dom = max(kernels, key=lambda k: kernels[k]["avg_us"] * kernels[k]["launches_per_step"])
        kd = kernels[dom]
        sum_kernel_us = sum(k["avg_us"] * k["launches_per_step"] for k in kernels.values())
        whole_bytes = vps * (877 * P + 156 * N + 48 * size * size) + 152 * P + 72 * F      # K0 runs once per step
        interactions = stats.get("interactions")
        # roofline of the dominant kernel, SURVEY.md 8(d): ALGORITHMIC bytes per launch / average launch duration (HIP events on
        # the launch stream, measured in this run) against 8 TB/s; `traffic` = HBM bytes from the PMC passes (null when the
        # committed counters belong to another build).  The compositing kernels have no dense contraction and gather 48-byte
        # records out of L2 -- what actually bounds them is VALU issue, reported next to it under `valu`.
        if kd.get("priced_with"):      # the dominant launch belongs to a multi-kernel stage: the stage is the roofline unit
            st = stages[kd["priced_with"]]
            kd = dict(kd, achieved_GBps=st["achieved_GBps"], frac_of_8TBps=st["frac_of_8TBps"], algorithmic_bytes=st["algorithmic_bytes"],
                      traffic=st["traffic"], avg_us=st["sum_us_per_view"])
            dom = kd["priced_with"]
        # `bound` / `achieved` / `peak` / `frac` are the HBM form SURVEY.md 8(d) prescribes for every kernel.  For the compositing kernels
        # that is NOT the resource that binds them (round-5 review, item 8): `binding_resource` names the one that does and `valu` prices it.
        roofline = {"kernel": dom, "avg_launch_us": kd["avg_us"], "bound": "hbm", "achieved": kd["achieved_GBps"], "peak": 8000.0,
                    "unit": "GB/s", "frac": kd["frac_of_8TBps"], "algorithmic_bytes": kd["algorithmic_bytes"],
                    "traffic": kd["traffic"], "pmc": pmc_note}
        if dom.startswith("blend"):
            vmodel, vnote = valu_model()
            vi = kd.get("valu_wave_insts")
            ach = vi * 64.0 / (kd["avg_us"] * 1e-6) / 1e12 if vi else None
            ns_mix = vmodel.get(dom, {}).get("ns_per_inst_mix_weighted")
            floor_us = vi * ns_mix * 1e-3 / 1024.0 if (vi and ns_mix) else None          # wave-instructions x ns each / 1 024 SIMDs
            roofline["binding_resource"] = ("valu_issue: the kernel issues %s vector wave-instructions per launch; at the issue costs measured on this "
                                            "chip they need %s us of its 1 024 SIMDs, %s of the launch -- against %s of the HBM roofline.  More resident "
                                            "waves do not shorten it (128-entry units at 8 blocks per CU: 126.2 us against 125.7, "
                                            "profiles/r06c_*), integer LDS atomics already run at the rate of stores (round 5)"
                                            % (f"{vi / 1e6:.1f} M" if vi else "?", f"{floor_us:.1f}" if floor_us else "?",
                                               f"{floor_us / kd['avg_us']:.2f}" if floor_us else "?", kd["frac_of_8TBps"])) if dom == "blend_bwd" else "valu_issue + staging latency"
            roofline["valu"] = {"achieved": round(ach, 2) if ach else None, "peak": round(VALU_PEAK_TLANEOPS, 1), "unit": "Tlane-op/s",
                                "frac": round(ach / VALU_PEAK_TLANEOPS, 4) if ach else None,
                                "wave_insts": vi, "ns_per_inst_mix_weighted": ns_mix,
                                "issue_time_floor_us": round(floor_us, 1) if floor_us else None,
                                "frac_mix_weighted": round(floor_us / kd["avg_us"], 4) if floor_us else None,
                                "mix": vmodel.get(dom, {}).get("walk_loop", {}).get("mix"), "model": vnote,
                                "interactions_sum_n_contrib": interactions,
                                "active_lane_frac": sq.get(dom, {}).get("active_lane_frac"),
                                "active_pairs": sq.get(dom, {}).get("active_pairs"),
                                "note": "vector lane-operations/s = SQ_INSTS_VALU x 64 / the HIP-event duration measured here; peak = 256 CU x 4 SIMD "
                                        "x 64 lanes / 1.2 ns, the issue cost of a plain f32 wave-instruction measured by tools/valu_bench.hip "
                                        "(DPP 1.76, packed / f64 1.8, transcendental 3.4 ns); `frac_mix_weighted` prices the walk loop's own "
                                        "instruction mix"}
        std = workload in ("c2_hotdog_like", "c4_ficus_like") and size == 800
        out = {
            "metric": BASELINE_METRIC if std else f"train iters/s (fwd+bwd raster) @{size}×{size}, {round(P / 1000)}k Gaussians; HBM GB/s vs roofline",
            "value": round(value, 2), "unit": "iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "repeats": {"n": len(regions), "ms_per_step": [round(1000.0 * r / args.steps, 4) for r in regions],
                        "median": round(ms_per_step, 4), "min": round(1000.0 * min(regions) / args.steps, 4),
                        "max": round(1000.0 * max(regions) / args.steps, 4),
                        "note": "every region: exactly --steps steps between barrier + synchronize on both sides; value = the median region"},
            "host_wait_us_per_step": round(host_wait_us_per_step, 1),
            "count_readback": ("deferred to the start of the backward (diff_gaussian_rasterization.set_deferred_counts); steps redone after an "
                               f"overflow: {deferred_overflows[0]}") if defer_counts else "inside the forward (blocking)",
            "config": {"workload": f"{desc['text']}, SH degree 3, {size}x{size}, orbit camera k=(rank*views+v)%8, white bg",
                       "model": desc["model"], "gaussians": P, "faces": F, "image": [size, size], "instances_N": N,
                       "interactions": interactions, "interactions_kind": stats.get("interactions_kind"),
                       "visible_gaussians": stats.get("visible"), "deepest_tile": stats.get("deepest_tile"),
                       "mean_instances_per_tile": round(N / max(1, ((size + 15) // 16) ** 2), 1),
                       "views_per_step": world * vps, "views_per_rank_per_step": vps,
                       "parallelism": (f"view-parallel x{world}, {vps} view(s) per rank per step, one gradient all-reduce "
                                       f"per step") if world > 1 else "single view",
                       "k0": ("inside the rasterizer's preprocess thread (update_alpha / prepare_scaling_rot deferred: games_hip.model.HipMeshMixin.hip_defer_k0)"
                              if getattr(model, "hip_defer_k0", False) else "its own launch"),
                       "k0_backward": ("inside preprocess_bwd (GmsRasterBackwardArgs.mesh, ABI 8): no mesh_bwd launch" if k0_bwd_inside
                                       else ("its own launch (mesh_bwd)" if ktimes.get("preprocess_bwd", (0, 0))[1] > 0 else "not profiled (--profile-steps 0)")),
                       "step": (f"K0 fwd + {vps} x (render fwd + bwd)" if vps > 1 else "K0 fwd + render fwd + bwd")
                               + (" + gradient all-reduce" if distributed else "")
                               + (" with the fused L1+SSIM training loss" if args.loss == "l1_ssim" else "")
                               + (f" + optimizer.step() [{args.optimizer}]" if args.optimizer != "none" else "")},
            "roofline": roofline,
            "kernels": kernels,
            "stages": stages,
            "kernel_timing": {"method": "HIP events on the launch stream around every launch of a separate untimed pass, minus the event "
                                        "pair's own overhead (gms_profile_event_overhead_us: a kernel that times itself by the device wall clock)",
                              "event_overhead_us": round(event_overhead_us, 2)},
            "whole_iteration": {"algorithmic_bytes": whole_bytes, "sum_kernel_us": round(sum_kernel_us, 1),
                                "achieved_GBps": round(whole_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                                "frac_of_8TBps": round(whole_bytes / (ms_per_step * 1e-3) / 8e12, 4)},
        }
        out.update(extra)
        if world == 1 and not args.no_cpu_baseline and desc["model"] == "gs_mesh":
            try:
                out["cpu_baseline"] = cpu_baseline(workload, args.state)
                out["cpu_baseline"]["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"] = {"error": repr(e)}
    if distributed:
        dist.destroy_process_group()
    if rank == 0:
        if distributed:
            out["rccl"] = rccl_summary(rccl_log)      # (after the communicator is gone: its debug file is complete)
        final_line = json.dumps(out)
        # RCCL prints its version banner through C stdio (flushed at exit): flush it first so that the JSON is the last line
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        print(final_line, flush=True)


if __name__ == "__main__":
    main()
