"""CPU analysis of the compositing workload (no GPU): for a synthetic scene, how many (splat, pixel-block) pairs
survive each candidate cull and how many lanes of a wave would carry an active pixel x splat pair.

    python tools/blend_stats.py [workload] [max_tiles]

Uses the CPU oracle (test infrastructure) only to obtain the per-tile sorted lists, conics and n_contrib; the
statistics themselves are numpy.  Output: one JSON object (pairs, survivors per cull, active-lane fractions) that
DESIGN.md quotes when choosing the blend kernels' wave mapping."""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (ROOT, os.path.join(ROOT, "gaussian-mesh-splatting_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

from games_hip import synthetic as syn
from oracle import gs_oracle, mesh_oracle

wl = sys.argv[1] if len(sys.argv) > 1 else "c2_hotdog_like"
max_tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 400
sc = syn.mesh_scene(wl, state="trained")
size = sc.meta["image"]
cam = syn.orbit_camera(0, width=size, height=size)
with torch.no_grad():
    _, _, xyz, scaling, rot = mesh_oracle.mesh_to_gaussians(sc.vertices, sc.faces, sc._alpha, sc._scale)
    cal = mesh_oracle.activated(xyz, scaling, rot, sc._opacity, sc._features_dc, sc._features_rest)
o = gs_oracle.rasterize(means3D=cal[0], opacities=cal[3], shs=cal[4], scales=cal[1], rotations=cal[2],
                        image_height=size, image_width=size, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.ones(3),
                        viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=3,
                        campos=cam.camera_center)
d = o.state.details()
xy, con, pl, ranges, ncon = d["xy"], d["conic_op"], d["point_list"], d["ranges"], d["n_contrib"]
gx = (size + 15) // 16
T = ranges.shape[0]
rng = np.random.default_rng(0)
nonempty = np.nonzero(ranges[:, 1] > ranges[:, 0])[0]
tiles = nonempty if len(nonempty) <= max_tiles else rng.choice(nonempty, max_tiles, replace=False)

A, B, Cc, op = con[:, 0], con[:, 1], con[:, 2], con[:, 3]
tau = np.log(255.0 * np.maximum(op, 1e-30)) + 1e-3
# cov from conic: inverse of [[A,B],[B,C]]
det = A * Cc - B * B
cxx, cyy = Cc / det, A / det
ext_x = np.sqrt(np.maximum(0, 2 * cxx * tau))
ext_y = np.sqrt(np.maximum(0, 2 * cyy * tau))

acc = dict(instances=0, pix_pairs_total=0, active_fwd=0, active_bwd=0)
blocks = {"16x16": (16, 16), "8x8": (8, 8), "8x4": (8, 4), "4x4": (4, 4), "16x4": (16, 4), "16x1": (16, 1), "4x2": (4, 2)}
st = {k: dict(bbox=0, exact=0, circ=0, anyact=0, act_lanes=0) for k in blocks}
# sub-wave streams: a wave owns an 8x8 quadrant; with S independent lane groups (S = 1: the whole quadrant, 2: two 8x4
# halves, 4: four 4x4 blocks) a trip advances every group by one entry of ITS OWN culled list, so a 64-entry batch
# costs max-over-groups trips instead of the size of the union list
streams = {"1x(8x8)": (8, 8), "2x(8x4)": (8, 4), "4x(4x4)": (4, 4)}
trips = {k: dict(exact=0, circ=0) for k in streams}


def circ_hit(a, b, c, cx, cy, x0, y0, x1, y1, thr):
    """conservative: centre distance in the splat's metric against sqrt(thr) + the metric radius of the block"""
    mx, my = 0.5 * (x0 + x1), 0.5 * (y0 + y1)
    hx, hy = 0.5 * (x1 - x0), 0.5 * (y1 - y0)
    dx, dy = mx - cx, my - cy
    qc = a * dx * dx + 2 * b * dx * dy + c * dy * dy
    rho = np.sqrt(a * hx * hx + c * hy * hy + 2 * np.abs(b) * hx * hy)
    return np.sqrt(qc) <= np.sqrt(thr) + rho


def qmin_rect(a, b, c, cx, cy, x0, y0, x1, y1):
    """min over the rectangle [x0,x1]x[y0,y1] of a dx^2 + 2 b dx dy + c dy^2, d = (x - cx, y - cy); arrays broadcast."""
    # clamp centre: if inside, 0
    px = np.clip(cx, x0, x1); py = np.clip(cy, y0, y1)
    inside = (px == cx) & (py == cy)
    best = np.full(np.broadcast(a, x0).shape, np.inf)
    # four edges: for fixed x = xe, minimise over y: y* = cy - b (xe-cx)/c clamped
    for xe in (x0, x1):
        dx = xe - cx
        ys = np.clip(cy - b * dx / c, y0, y1); dy = ys - cy
        best = np.minimum(best, a * dx * dx + 2 * b * dx * dy + c * dy * dy)
    for ye in (y0, y1):
        dy = ye - cy
        xs = np.clip(cx - b * dy / a, x0, x1); dx = xs - cx
        best = np.minimum(best, a * dx * dx + 2 * b * dx * dy + c * dy * dy)
    return np.where(inside, 0.0, best)


for t in tiles:
    lo, hi = ranges[t]
    ids = pl[lo:hi]
    n = len(ids)
    tx, ty = t % gx, t // gx
    px = (tx * 16 + np.arange(16))[None, None, :].astype(np.float64)
    py = (ty * 16 + np.arange(16))[None, :, None].astype(np.float64)
    a, b, c = A[ids][:, None, None], B[ids][:, None, None], Cc[ids][:, None, None]
    cx, cy = xy[ids, 0][:, None, None], xy[ids, 1][:, None, None]
    dx, dy = cx - px, cy - py
    pw = -0.5 * (a * dx * dx + c * dy * dy) - b * dx * dy
    al = np.minimum(0.99, op[ids][:, None, None] * np.exp(pw))
    passes = (pw <= 0) & (al >= 1.0 / 255.0)                                  # [n,16,16]
    nc = ncon[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16]
    pos = np.arange(n)[:, None, None]
    within = pos < nc[None, :nc.shape[0], :nc.shape[1]] if nc.shape == (16, 16) else None
    if within is None:
        continue
    act = passes & within          # pairs the backward processes (and the forward applies, up to the stop pixel)
    acc["instances"] += n
    acc["pix_pairs_total"] += n * 256
    acc["active_bwd"] += int(act.sum())
    for name, (bw, bh) in blocks.items():
        nbx, nby = 16 // bw, 16 // bh
        s = st[name]
        for by in range(nby):
            for bx in range(nbx):
                x0, y0 = tx * 16 + bx * bw, ty * 16 + by * bh
                x1, y1 = x0 + bw - 1, y0 + bh - 1
                bb = ~((xy[ids, 0] + ext_x[ids] < x0) | (xy[ids, 0] - ext_x[ids] > x1) | (xy[ids, 1] + ext_y[ids] < y0) |
                       (xy[ids, 1] - ext_y[ids] > y1))
                qm = qmin_rect(A[ids], B[ids], Cc[ids], xy[ids, 0], xy[ids, 1], x0, y0, x1, y1)
                ex = bb & (qm <= 2 * tau[ids])
                ci = bb & circ_hit(A[ids], B[ids], Cc[ids], xy[ids, 0], xy[ids, 1], x0, y0, x1, y1, 2 * tau[ids])
                blk = act[:, by * bh:(by + 1) * bh, bx * bw:(bx + 1) * bw].reshape(n, -1)
                # the block stops being walked after its furthest n_contrib
                live = pos[:, 0, 0] < nc[by * bh:(by + 1) * bh, bx * bw:(bx + 1) * bw].max()
                s["bbox"] += int((bb & live).sum()); s["exact"] += int((ex & live).sum())
                s["circ"] += int((ci & live).sum())
                s["anyact"] += int(blk.any(axis=1).sum()); s["act_lanes"] += int(blk.sum())
    for name, (bw, bh) in streams.items():
        for qy in range(2):
            for qx in range(2):
                cnt_e, cnt_c = [], []
                qlive = pos[:, 0, 0] < nc[qy * 8:qy * 8 + 8, qx * 8:qx * 8 + 8].max()
                for by in range(8 // bh):
                    for bx in range(8 // bw):
                        x0, y0 = tx * 16 + qx * 8 + bx * bw, ty * 16 + qy * 8 + by * bh
                        x1, y1 = x0 + bw - 1, y0 + bh - 1
                        bb = ~((xy[ids, 0] + ext_x[ids] < x0) | (xy[ids, 0] - ext_x[ids] > x1) |
                               (xy[ids, 1] + ext_y[ids] < y0) | (xy[ids, 1] - ext_y[ids] > y1)) & qlive
                        qm = qmin_rect(A[ids], B[ids], Cc[ids], xy[ids, 0], xy[ids, 1], x0, y0, x1, y1)
                        cnt_e.append(bb & (qm <= 2 * tau[ids]))
                        cnt_c.append(bb & circ_hit(A[ids], B[ids], Cc[ids], xy[ids, 0], xy[ids, 1], x0, y0, x1, y1, 2 * tau[ids]))
                for key, lst in (("exact", cnt_e), ("circ", cnt_c)):
                    m = np.stack(lst).astype(np.int64)                       # [groups, n]
                    pad = (-n) % 64
                    m = np.pad(m, ((0, 0), (0, pad))).reshape(m.shape[0], -1, 64).sum(axis=2)    # per 64-entry batch
                    trips[name][key] += int(m.max(axis=0).sum())

out = {"workload": wl, "tiles_sampled": int(len(tiles)), "tiles_nonempty": int(len(nonempty)), "N_total": int(d["N"]),
       "instances_sampled": acc["instances"], "active_pairs_sampled": acc["active_bwd"],
       "active_pairs_per_instance": round(acc["active_bwd"] / max(1, acc["instances"]), 2), "blocks": {}}
for name, (bw, bh) in blocks.items():
    s = st[name]
    lanes = bw * bh
    out["blocks"][name] = {
        "pairs_bbox_per_instance": round(s["bbox"] / acc["instances"], 3),
        "pairs_exact_per_instance": round(s["exact"] / acc["instances"], 3),
        "pairs_circle_per_instance": round(s["circ"] / acc["instances"], 3),
        "pairs_anyactive_per_instance": round(s["anyact"] / acc["instances"], 3),
        "exact_over_bbox": round(s["exact"] / max(1, s["bbox"]), 3),
        "lane_evals_bbox_per_instance": round(s["bbox"] * lanes / acc["instances"], 1),
        "lane_evals_exact_per_instance": round(s["exact"] * lanes / acc["instances"], 1),
        "active_lane_frac_bbox": round(s["act_lanes"] / max(1, s["bbox"] * lanes), 3),
        "active_lane_frac_exact": round(s["act_lanes"] / max(1, s["exact"] * lanes), 3)}
out["wave_trips_per_instance"] = {k: {c: round(v[c] / acc["instances"], 3) for c in v} for k, v in trips.items()}
print(json.dumps(out, indent=1))
