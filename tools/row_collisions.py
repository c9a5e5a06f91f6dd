"""CPU statistic (no GPU) for the micro-tile backward: how often do the four rows of a wave add into the SAME entry of the
unit's LDS gradient table in the same trip?  Rows = the unit's sixteen 4x4 blocks ordered by list length, four to a wave,
aligned at their tops (blend_micro.hip::micro_bwd_kernel).  Same-address LDS atomics serialise.

    python tools/row_collisions.py [workload] [max_tiles]"""
import json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (ROOT, os.path.join(ROOT, "gaussian-mesh-splatting_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from games_hip import synthetic as syn
from oracle import gs_oracle, mesh_oracle

wl = sys.argv[1] if len(sys.argv) > 1 else "c2_hotdog_like"
max_tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 150
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
tau = np.log(255.0 * np.maximum(op, 1e-30)) + 1e-3
tot_rows = tot_distinct = trips = 0
mult_hist = np.zeros(5, np.int64)
spatial = dict(rows=0, distinct=0)
for t in tiles:
    lo, hi = ranges[t]
    ids = pl[lo:hi]; n = len(ids)
    tx, ty = t % gx, t // gx
    nc = ncon[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16]
    if nc.shape != (16, 16):
        continue
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
        for order_name in ("length", "spatial"):
            order = np.argsort([-len(x) for x in lists], kind="stable") if order_name == "length" else np.array([0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15])
            for w in range(4):
                rows = order[4 * w:4 * w + 4]
                mt = max(tops[r] for r in rows)
                for k in range(mt):
                    es = [lists[r][tops[r] - 1 - k] for r in rows if k < tops[r]]
                    if order_name == "length":
                        tot_rows += len(es); tot_distinct += len(set(es)); trips += 1
                        mult_hist[max(np.unique(es, return_counts=True)[1])] += 1
                    else:
                        spatial["rows"] += len(es); spatial["distinct"] += len(set(es))
print(json.dumps(dict(workload=wl, tiles=int(len(tiles)), trips=int(trips), row_adds=int(tot_rows), distinct_addresses=int(tot_distinct),
                      same_address_frac=round(1 - tot_distinct / max(1, tot_rows), 4),
                      max_multiplicity_hist={str(i): int(mult_hist[i]) for i in range(1, 5)},
                      spatial_grouping_same_address_frac=round(1 - spatial["distinct"] / max(1, spatial["rows"]), 4))))
