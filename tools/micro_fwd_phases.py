"""Where the waves of the micro-tile FORWARD launches spend their time (round 6).  EXPERIMENTS build (tools/build_experiments.sh);
on the GPU box:
    LD_LIBRARY_PATH=.../lib_exp GMSPLAT_LIB=.../lib_exp/libgmsplat.so GMS_DBG=2048 python tools/micro_fwd_phases.py      # micro_head
    ... GMS_DBG=4096 ...                                                                                               # micro_fwd
Stamps per wave (100 MHz wall clock): 0 start, 1 staged (ids, records, masks / filter, lists), 5 end; word 6 = kind of unit
(1 first segment exact walk, 2 transmittance products of a middle segment, 3 last segment, 4 middle segment exact walk)."""
import ctypes as C, os, sys
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
with torch.no_grad():
    for it in range(4):
        model.update_alpha(); model.prepare_scaling_rot()
        img = render(cam, model, PipelineParams(), bg)["render"]
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(4 * 65536 * 8, np.uint64)
lib.gms_debug_read.argtypes = [C.c_void_p, C.c_size_t]
print("gms_debug_read rc", lib.gms_debug_read(buf.ctypes.data_as(C.c_void_p), buf.nbytes), "GMS_DBG", os.environ.get("GMS_DBG"))
b = buf.reshape(4, 65536, 8).astype(np.int64)
live = (b[:, :, 0] > 0) & (b[:, :, 5] > 0)
if not live.any():
    sys.exit("no stamps recorded")
t0 = b[:, :, 0][live].min()
st, mid, en = [(b[:, :, k] - t0) / 100.0 for k in (0, 1, 5)]
kind = b[:, :, 6]
print(f"{int(live.sum())} waves in {int(live.any(axis=0).sum())} blocks; kernel span {en[live].max():.1f} us; "
      f"sum of wave lifetimes {(en - st)[live].sum():.0f} wave-us = {(en - st)[live].sum() / 8192:.1f} us of a chip with all 8192 wave slots busy")
names = {1: "first segment, exact walk", 2: "middle segment, products", 3: "last segment, exact walk", 4: "middle segment, exact walk"}
for k, n in names.items():
    m = live & (kind == k)
    if not m.any():
        continue
    a, w = (mid - st)[m], (en - mid)[m]
    print(f"  {n:30s} {int(m.sum()) // 4:5d} blocks  staging mean {a.mean():5.2f} p90 {np.percentile(a, 90):5.2f}   walk mean {w.mean():5.2f} p90 {np.percentile(w, 90):5.2f} max {w.max():5.2f} us"
          f"   share of wave-time: staging {100 * a.sum() / (en - st)[live].sum():4.1f} % walk {100 * w.sum() / (en - st)[live].sum():4.1f} %")
if (b[:, :, 3][live] > 0).all() and (b[:, :, 4][live] > 0).all():
    # inside staging: 0 -> 3 unit record, key, splat record (three dependent round trips) + block mask; 3 -> 4 ballots, counts to LDS, barrier;
    # 4 -> 1 list bytes, block order, barrier
    r_in, xch = [(b[:, :, k] - t0) / 100.0 for k in (3, 4)]
    for nm, lo, hi in (("records in", st, r_in), ("counts exchanged (barrier)", r_in, xch), ("lists built (barrier)", xch, mid)):
        d = (hi - lo)[live]
        print(f"    staging / {nm:28s} mean {d.mean():5.2f} p50 {np.percentile(d, 50):5.2f} p90 {np.percentile(d, 90):5.2f} max {d.max():5.2f} us")
    early = live & (st < 2.0)
    if early.any():
        d = (r_in - st)[early]
        print(f"    waves started in the first 2 us ({int(early.sum())}): records in after mean {d.mean():5.2f} p90 {np.percentile(d, 90):5.2f} us")
blk = np.nonzero(live.all(axis=0))[0]
if os.environ.get("GMS_PHASES_DUMP"):          # per-block records for offline scheduling studies (tools/unit_order_study.py)
    w7 = b[0, blk, 7]
    np.savez_compressed(os.environ["GMS_PHASES_DUMP"], block=blk, kind=kind[0, blk], start=st[:, blk].min(axis=0), staged=mid[:, blk].max(axis=0),
                        end=en[:, blk].max(axis=0), unit=(w7 >> 40) & 0xffffff, entries=(w7 >> 28) & 0xfff, nseg=(w7 >> 14) & 0x3fff, seg=w7 & 0x3fff,
                        tile_entries=b[0, blk, 2])
bd = en[:, blk].max(axis=0) - st[:, blk].min(axis=0)
wk = (en - mid)[:, blk]
print(f"blocks: duration mean {bd.mean():.1f} p50 {np.percentile(bd, 50):.1f} p90 {np.percentile(bd, 90):.1f} max {bd.max():.1f} us; "
      f"longest walk of a block / mean walk of its waves = {wk.max(axis=0).sum() * 4 / max(wk.sum(), 1e-9):.2f}")
ts = np.linspace(0, en[live].max(), 24)
s_, e_, m_ = st[live], en[live], mid[live]
print("resident waves over time :", [int(((s_ <= x) & (e_ > x)).sum()) for x in ts])
print("... of which walking     :", [int(((m_ <= x) & (e_ > x)).sum()) for x in ts])
# the stragglers: the blocks that end last -- when they started, how long they staged and walked, what they are
bend = en[:, blk].max(axis=0); bst = st[:, blk].min(axis=0); bmid = mid[:, blk].max(axis=0)
order = np.argsort(-bend)[:24]
print("last blocks to end (block index: start -> staged -> end us, kind):")
print("  " + "  ".join(f"{int(blk[i])}: {bst[i]:.1f}->{bmid[i]:.1f}->{bend[i]:.1f} k{int(kind[0, blk[i]])}" for i in order))
for frac in (0.5, 0.8, 0.9, 0.95, 0.99):
    print(f"  {100 * frac:.0f} % of the blocks have ended by {np.quantile(bend, frac):.1f} us", end=";")
print()
late = bend > np.quantile(bend, 0.97)
print(f"the last 3 % of blocks: start mean {bst[late].mean():.1f} (all: {bst.mean():.1f}), staging mean {(bmid - bst)[late].mean():.1f} (all {(bmid - bst).mean():.1f}), "
      f"walk mean {(bend - bmid)[late].mean():.1f} (all {(bend - bmid).mean():.1f}), block index mean {blk[late].mean():.0f} of {blk.max()}")
