"""CPU statistic (no GPU) for the micro-tile backward: wave trips when the unit's sixteen 4x4 blocks are dealt four to a wave by LIST LENGTH
(what unit_stage does, for every launch) against by the number of entries the block's pixels actually composited (known in the backward
from n_contrib).  A wave walks max(top of its four rows) positions, two per trip.      python tools/row_order_study.py [workload] [max_tiles]"""
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
trips = dict(length=0, top=0, ideal_pairs=0)
units = 0
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
    hit = (pw <= 0) & (op[ids][:, None, None] * np.exp(pw) >= 1.0 / 255.0)
    for s0 in range(0, n, L):
        s1 = min(n, s0 + L)
        lens, tops = [], []
        for by in range(4):
            for bx in range(4):
                h = hit[s0:s1, 4 * by:4 * by + 4, 4 * bx:4 * bx + 4].reshape(s1 - s0, -1).any(axis=1)
                lst = np.nonzero(h)[0]
                last = int(nc[4 * by:4 * by + 4, 4 * bx:4 * bx + 4].max())
                lens.append(len(lst)); tops.append(int((lst + s0 < last).sum()))
        if max(tops) == 0:
            continue
        units += 1
        lens, tops = np.array(lens), np.array(tops)
        for name, key in (("length", -lens), ("top", -tops)):
            order = np.argsort(key, kind="stable")
            trips[name] += sum((int(tops[order[4 * w:4 * w + 4]].max()) + 1) // 2 for w in range(4))
        trips["ideal_pairs"] += (int(tops.sum()) + 7) // 8          # four rows x two positions per trip, perfectly packed
print(json.dumps(dict(workload=wl, tiles=int(len(tiles)), units=units, wave_trips_dealt_by_list_length=trips["length"], wave_trips_dealt_by_composited_depth=trips["top"],
                      ratio=round(trips["top"] / max(1, trips["length"]), 4), perfectly_packed=trips["ideal_pairs"])))
