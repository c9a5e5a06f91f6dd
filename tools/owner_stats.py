"""CPU statistic (no GPU) for the micro-tile backward: which share of the per-(4x4 block, splat) gradient sums could be added to
the unit's LDS table WITHOUT an atomic?  An LDS float atomic costs ~2.7 cycles per active lane on MI355X (tools/lds_bench.hip:
45 ns per 40-lane ds_add_f32 per CU against 3.4 ns for a plain read-add-write), and the backward issues 16.6 M lane-adds per frame:
~75 us of a serial per-CU resource.  An entry of the unit that only ONE wave of the block ever touches can take a plain
read-add-write by that wave.  Counted here, for the two ways of dealing the sixteen blocks to the four waves (by list length, as
shipped; by 8x8 quadrant): row-adds whose entry is single-wave, trips, and rows of one trip that name the same entry.

    python tools/owner_stats.py [workload] [max_tiles]"""
import json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (ROOT, os.path.join(ROOT, "gaussian-mesh-splatting_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from games_hip import synthetic as syn
from oracle import gs_oracle, mesh_oracle

wl = sys.argv[1] if len(sys.argv) > 1 else "c2_hotdog_like"
max_tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 200
L = 256
sc = syn.mesh_scene(wl, state="trained")
size = sc.meta["image"]
cam = syn.orbit_camera(0, width=size, height=size)
with torch.no_grad():
    _, _, xyz, scaling, rot = mesh_oracle.mesh_to_gaussians(sc.vertices, sc.faces, sc._alpha, sc._scale)
    cal = mesh_oracle.activated(xyz, scaling, rot, sc._opacity, sc._features_dc, sc._features_rest)
o = gs_oracle.rasterize(means3D=cal[0], opacities=cal[3], shs=cal[4], scales=cal[1], rotations=cal[2], image_height=size, image_width=size,
                        tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.ones(3), viewmatrix=cam.world_view_transform,
                        projmatrix=cam.full_proj_transform, sh_degree=3, campos=cam.camera_center)
d = o.state.details()
xy, con, pl, ranges, ncon = d["xy"], d["conic_op"], d["point_list"], d["ranges"], d["n_contrib"]
gx = (size + 15) // 16
rng = np.random.default_rng(0)
nonempty = np.nonzero(ranges[:, 1] > ranges[:, 0])[0]
tiles = nonempty if len(nonempty) <= max_tiles else rng.choice(nonempty, max_tiles, replace=False)
A, B, Cc, op = con[:, 0], con[:, 1], con[:, 2], con[:, 3]
SPATIAL = np.array([0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15])
st = {k: dict(adds=0, single=0, trips=0, dup_rows=0, units=0, single_entries=0, entries=0) for k in ("length", "spatial")}
inst = 0
for t in tiles:
    lo, hi = ranges[t]
    ids = pl[lo:hi]; n = len(ids)
    tx, ty = t % gx, t // gx
    nc = ncon[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16]
    if nc.shape != (16, 16):
        continue
    inst += n
    px = (tx * 16 + np.arange(16))[None, None, :].astype(np.float64); py = (ty * 16 + np.arange(16))[None, :, None].astype(np.float64)
    a, b, c = A[ids][:, None, None], B[ids][:, None, None], Cc[ids][:, None, None]
    dx, dy = xy[ids, 0][:, None, None] - px, xy[ids, 1][:, None, None] - py
    pw = -0.5 * (a * dx * dx + c * dy * dy) - b * dx * dy
    hit = (pw <= 0) & (op[ids][:, None, None] * np.exp(pw) >= 1.0 / 255.0)         # exact per-pixel test: a lower bound of the filter
    for s0 in range(0, n, L):
        s1 = min(n, s0 + L)
        lists, tops = [], []
        for by in range(4):
            for bx in range(4):
                h = hit[s0:s1, 4 * by:4 * by + 4, 4 * bx:4 * bx + 4].reshape(s1 - s0, -1).any(axis=1)
                lst = np.nonzero(h)[0]
                last = int(nc[4 * by:4 * by + 4, 4 * bx:4 * bx + 4].max())
                lists.append(lst); tops.append(int((lst + s0 < last).sum()))
        for name in ("length", "spatial"):
            S = st[name]
            order = np.argsort([-len(x) for x in lists], kind="stable") if name == "length" else SPATIAL
            wave_of = np.empty(16, int); wave_of[order] = np.arange(16) // 4
            touched = {}                                   # entry -> set of waves whose rows reach it in the backward walk
            for bidx in range(16):
                for e in lists[bidx][:tops[bidx]]:
                    touched.setdefault(int(e), set()).add(int(wave_of[bidx]))
            S["units"] += 1; S["entries"] += len(touched); S["single_entries"] += sum(1 for v in touched.values() if len(v) == 1)
            for w in range(4):
                rows = order[4 * w:4 * w + 4]
                mt = max(tops[r] for r in rows)
                S["trips"] += mt
                for k in range(mt):
                    es = [int(lists[r][k]) for r in rows if k < tops[r]]          # rows aligned at the bottom: list position k
                    S["adds"] += len(es)
                    S["single"] += sum(1 for e in es if len(touched[e]) == 1)
                    S["dup_rows"] += len(es) - len(set(es))
out = dict(workload=wl, tiles=int(len(tiles)), instances=int(inst))
for name, S in st.items():
    out[name] = dict(trips_per_instance=round(S["trips"] / max(1, inst), 3), row_adds=S["adds"],
                     single_wave_add_frac=round(S["single"] / max(1, S["adds"]), 4),
                     single_wave_entry_frac=round(S["single_entries"] / max(1, S["entries"]), 4),
                     same_entry_rows_per_add=round(S["dup_rows"] / max(1, S["adds"]), 4))
print(json.dumps(out, indent=1))
