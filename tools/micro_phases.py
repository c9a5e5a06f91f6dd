"""Where the waves of the micro-tile backward spend their time ("measure where the parked cycles go").

Needs a library built with `make EXPERIMENTS=1` (tools/build_experiments.sh puts one under lib_exp/); run on the GPU box:

    LD_LIBRARY_PATH=gaussian-mesh-splatting_amd/lib_exp GMSPLAT_LIB=gaussian-mesh-splatting_amd/lib_exp/libgmsplat.so \
        GMS_DBG=1024 [GMS_MICRO_RU=0|1] python tools/micro_phases.py [workload]

Every wave of `micro_bwd` / `ru_bwd` stamps the 100 MHz wall clock at: 0 start, 1 after the unit staging (ids / records / lists,
its barriers), 2 before the walk (pixel state, segment restart), 3 after the walk, 4 after the block barrier, 5 at the end (flush
issued); word 6 = the wave's trip count (the longest of its four rows).  Printed: the share of all wave-time in each phase, the
block-level view (how long the block's other waves wait for the slowest one) and the kernel span.
"""
import ctypes as C, os, sys
os.environ.setdefault("GMS_DBG", "1024")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-mesh-splatting_amd"))
import numpy as np, torch
from diff_gaussian_rasterization import _lib
from games_hip import synthetic as syn
from games_hip.model import HipGaussianMeshModel
from games_hip.render import PipelineParams, render

wl = sys.argv[1] if len(sys.argv) > 1 else "c2_hotdog_like"
scene = syn.mesh_scene(wl, state="trained")
size = scene.meta["image"]
model = HipGaussianMeshModel.from_scene(scene, "cuda")
cam = syn.orbit_camera(0, width=size, height=size).to("cuda"); bg = torch.ones(3, device="cuda")
for it in range(4):
    model.update_alpha(); model.prepare_scaling_rot()
    img = render(cam, model, PipelineParams(), bg)["render"]
    img.backward((img.detach() - 0.5) / img.numel())
    for p in model.parameters(): p.grad = None
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(4 * 65536 * 8, np.uint64)
lib.gms_debug_read.argtypes = [C.c_void_p, C.c_size_t]
rc = lib.gms_debug_read(buf.ctypes.data_as(C.c_void_p), buf.nbytes)
print("gms_debug_read rc", rc, "(-1: not an EXPERIMENTS build or GMS_DBG & 1024 unset)")
b = buf.reshape(4, 65536, 8).astype(np.int64)
live = (b[:, :, 0] > 0) & (b[:, :, 5] > 0)          # waves that ran to the end (blocks without a unit leave early)
if not live.any():
    sys.exit("no stamps recorded")
t = b[:, :, :6].astype(np.float64)
t[:, :, 2] = np.where(b[:, :, 2] > 0, t[:, :, 2], t[:, :, 1])          # a wave with nothing to walk skips stamp 2
t0 = t[:, :, 0][live].min()
t = (t - t0) / 100.0                                   # microseconds
trips = b[:, :, 6]
names = ["staging (ids, records, lists; 2 barriers)", "prologue (pixel state, segment restart)", "walk", "barrier wait", "flush"]
d = np.stack([t[:, :, k + 1] - t[:, :, k] for k in range(5)], axis=-1)
tot = d[live].sum()
print(f"workload {wl}: {int(live.sum())} waves in {int(live.any(axis=0).sum())} blocks; kernel span {t[:, :, 5][live].max():.1f} us; "
      f"sum of wave lifetimes {tot:.0f} wave-us = {tot / 8192:.1f} us of a chip with all 8192 wave slots busy")
for k, n in enumerate(names):
    x = d[:, :, k][live]
    print(f"  {n:48s} {100 * x.sum() / tot:5.1f} % of wave-time   mean {x.mean():6.2f} us  p50 {np.percentile(x, 50):6.2f}  p90 {np.percentile(x, 90):6.2f}  max {x.max():6.2f}")
tr = trips[live]
walk = d[:, :, 2][live]
nz = tr > 0
print(f"trips per wave: mean {tr.mean():.1f} p50 {np.percentile(tr, 50):.0f} p90 {np.percentile(tr, 90):.0f} max {tr.max()}; "
      f"walk time per trip {1e3 * walk[nz].sum() / tr[nz].sum():.0f} ns = {2.4 * 1e3 * walk[nz].sum() / tr[nz].sum():.0f} cycles at 2.4 GHz")
blk = np.nonzero(live.all(axis=0))[0]
if os.environ.get("GMS_PHASES_DUMP"):          # per-block records for offline scheduling studies (tools/unit_order_study.py)
    w7 = b[0, blk, 7]
    np.savez_compressed(os.environ["GMS_PHASES_DUMP"], block=blk, kind=np.full(len(blk), 5), start=t[:, blk, 0].min(axis=0), staged=t[:, blk, 2].max(axis=0),
                        end=t[:, blk, 5].max(axis=0), unit=(w7 >> 40) & 0xffffff, entries=(w7 >> 28) & 0xfff, nseg=(w7 >> 14) & 0x3fff, seg=w7 & 0x3fff,
                        tile_entries=np.zeros(len(blk)), trips=trips[:, blk].max(axis=0), trips_sum=trips[:, blk].sum(axis=0))
bt = t[:, blk, :]
bdur = bt[:, :, 5].max(axis=0) - bt[:, :, 0].min(axis=0)
wmax = (bt[:, :, 3] - bt[:, :, 2]).max(axis=0)
wsum = (bt[:, :, 3] - bt[:, :, 2]).sum(axis=0)
print(f"blocks: duration mean {bdur.mean():.1f} us p50 {np.percentile(bdur, 50):.1f} p90 {np.percentile(bdur, 90):.1f} max {bdur.max():.1f}; "
      f"longest walk of a block / mean walk of its waves = {(wmax.sum() * 4 / max(wsum.sum(), 1e-9)):.2f}")
ts = np.linspace(0, t[:, :, 5][live].max(), 24)
s_, e_ = t[:, :, 0][live], t[:, :, 5][live]
w2, w3 = t[:, :, 2][live], t[:, :, 3][live]
print("resident waves over time :", [int(((s_ <= x) & (e_ > x)).sum()) for x in ts])
print("... of which in the walk :", [int(((w2 <= x) & (w3 > x)).sum()) for x in ts])
