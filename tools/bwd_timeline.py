"""Debug: per-wave start/end timeline of blend_bwd (GMS_DBG=16). Run on the GPU box."""
import ctypes as C, os, sys
os.environ["GMS_DBG"] = "16"
sys.path.insert(0, "."); sys.path.insert(0, "gaussian-mesh-splatting_amd")
import numpy as np, torch
from diff_gaussian_rasterization import _lib
from games_hip import synthetic as syn
from games_hip.model import HipGaussianMeshModel
from games_hip.render import PipelineParams, render
scene = syn.mesh_scene("c2_hotdog_like", state="trained")
model = HipGaussianMeshModel.from_scene(scene, "cuda")
cam = syn.orbit_camera(0).to("cuda"); bg = torch.ones(3, device="cuda")
for it in range(3):
    model.update_alpha(); model.prepare_scaling_rot()
    img = render(cam, model, PipelineParams(), bg)["render"]
    img.backward((img.detach() - 0.5) / img.numel())
    for p in model.parameters(): p.grad = None
torch.cuda.synchronize()
lib = _lib.load()
n = 4 * 65536 * 2
buf = np.zeros(n, np.uint64)
lib.gms_debug_read.argtypes = [C.c_void_p, C.c_size_t]
print("rc", lib.gms_debug_read(buf.ctypes.data_as(C.c_void_p), buf.nbytes))
b = buf.reshape(4, 65536, 2).astype(np.int64)
live = b[:, :, 0] > 0
t0 = b[:, :, 0][live].min()
start = (b[:, :, 0] - t0) / 100.0   # us (100 MHz)
end = (b[:, :, 1] - t0) / 100.0
dur = (end - start)[live]
print("waves recorded", live.sum(), "kernel span us", end[live].max())
print("wave duration us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max()))
print("start time us: p50 %.1f p90 %.1f p99 %.1f max %.1f" % (*np.percentile(start[live], [50, 90, 99]), start[live].max()))
# residency over time
ts = np.linspace(0, end[live].max(), 30)
s_, e_ = start[live], end[live]
print("resident waves over time:", [int(((s_ <= t) & (e_ > t)).sum()) for t in ts])
long_ = dur > np.percentile(dur, 99)
print("long waves: start p50 %.1f, dur mean %.1f" % (np.median(s_[long_]), dur[long_].mean()))
