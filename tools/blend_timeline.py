"""Debug: per-wave start/end timeline of one blend kernel.  GMS_DBG bit: 16 = blend_bwd, 128 = blend_fwd (segments >= 1),
256 = blend_head (first segments + transmittance products).  Needs a library built with `make EXPERIMENTS=1`.  Run on the GPU box:  python tools/blend_timeline.py 128"""
import ctypes as C, os, sys
bit = int(sys.argv[1]) if len(sys.argv) > 1 else 16
os.environ["GMS_DBG"] = str(bit)
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-mesh-splatting_amd"))
import numpy as np, torch
from diff_gaussian_rasterization import _lib
from games_hip import synthetic as syn
from games_hip.model import HipGaussianMeshModel
from games_hip.render import PipelineParams, render
wl = sys.argv[2] if len(sys.argv) > 2 else "c2_hotdog_like"
scene = syn.mesh_scene(wl, state="trained")
size = scene.meta["image"]
model = HipGaussianMeshModel.from_scene(scene, "cuda")
cam = syn.orbit_camera(0, width=size, height=size).to("cuda"); bg = torch.ones(3, device="cuda")
for it in range(3):
    model.update_alpha(); model.prepare_scaling_rot()
    img = render(cam, model, PipelineParams(), bg)["render"]
    img.backward((img.detach() - 0.5) / img.numel())
    for p in model.parameters(): p.grad = None
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(4 * 65536 * 2, np.uint64)
lib.gms_debug_read.argtypes = [C.c_void_p, C.c_size_t]
print("rc", lib.gms_debug_read(buf.ctypes.data_as(C.c_void_p), buf.nbytes))
b = buf.reshape(4, 65536, 2).astype(np.int64)
live = b[:, :, 0] > 0
t0 = b[:, :, 0][live].min()
start = (b[:, :, 0] - t0) / 100.0   # us (100 MHz)
end = (b[:, :, 1] - t0) / 100.0
dur = (end - start)[live]
print("waves recorded", int(live.sum()), "kernel span us %.1f" % end[live].max())
print("wave duration us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max()))
print("start time us: p50 %.1f p90 %.1f p99 %.1f max %.1f" % (*np.percentile(start[live], [50, 90, 99]), start[live].max()))
print("end time us: p50 %.1f p90 %.1f p99 %.1f max %.1f" % (*np.percentile(end[live], [50, 90, 99]), end[live].max()))
ts = np.linspace(0, end[live].max(), 30)
s_, e_ = start[live], end[live]
print("resident waves over time:", [int(((s_ <= t) & (e_ > t)).sum()) for t in ts])
print("sum of wave durations (wave-us): %.0f  => %.1f us of a fully occupied chip (8192 wave slots)" % (dur.sum(), dur.sum() / 8192))
blk = np.nonzero(live.any(axis=0))[0]
bd = (end[:, blk].max(axis=0) - np.where(live[:, blk], start[:, blk], 1e18).min(axis=0))
order = np.argsort(-bd)[:12]
print("longest blocks (index, start, duration):", [(int(blk[i]), round(float(np.where(live[:, blk[i]], start[:, blk[i]], 1e18).min()), 1), round(float(bd[i]), 1)) for i in order])
late = np.argsort(-end[:, blk].max(axis=0))[:12]
print("last-finishing blocks (index, start, end):", [(int(blk[i]), round(float(np.where(live[:, blk[i]], start[:, blk[i]], 1e18).min()), 1), round(float(end[:, blk[i]].max()), 1)) for i in late])
