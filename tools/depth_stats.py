"""CPU analysis (no GPU): how much of each tile's depth-sorted list is ever composited.

    python tools/depth_stats.py [workload]

For a synthetic scene the C oracle (test infrastructure) gives the per-tile lists and n_contrib; the deepest position any
pixel of a tile composites (max n_contrib over its 256 pixels) is the part of the list that has to be SORTED and WALKED at
all -- everything behind it is invisible to every pixel of the tile.  Output: one JSON object that DESIGN.md section 9 quotes
for the depth-bucketed binning of deep scenes."""
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (ROOT, os.path.join(ROOT, "gaussian-mesh-splatting_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

from games_hip import synthetic as syn
from oracle import gs_oracle, mesh_oracle

wl = sys.argv[1] if len(sys.argv) > 1 else "c5_flame_like_1m"
sc = syn.mesh_scene(wl, state="trained")
size = sc.meta["image"]
cam = syn.orbit_camera(0, width=size, height=size)
t0 = time.time()
with torch.no_grad():
    _, _, xyz, scaling, rot = mesh_oracle.mesh_to_gaussians(sc.vertices, sc.faces, sc._alpha, sc._scale)
    cal = mesh_oracle.activated(xyz, scaling, rot, sc._opacity, sc._features_dc, sc._features_rest)
o = gs_oracle.rasterize(means3D=cal[0], opacities=cal[3], shs=cal[4], scales=cal[1], rotations=cal[2], image_height=size,
                        image_width=size, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.ones(3),
                        viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=3,
                        campos=cam.camera_center)
d = o.state.details()
rg, nc = d["ranges"], d["n_contrib"]
gx = (size + 15) // 16
T = rg.shape[0]
length = (rg[:, 1] - rg[:, 0]).astype(np.int64)
H, W = nc.shape
pad = np.zeros(((H + 15) // 16 * 16, (W + 15) // 16 * 16), nc.dtype)
pad[:H, :W] = nc
need = pad.reshape(pad.shape[0] // 16, 16, pad.shape[1] // 16, 16).max(axis=(1, 3)).reshape(-1).astype(np.int64)[:T]
nz = length > 0
q = [50, 90, 99, 100]
# What a pipeline can REALISE without knowing the answer: `need` is the oracle's hindsight (a pixel that never saturates still has a
# last contributor, but only a walk to the end of the list proves it).  A tile's tail can be skipped -- unsorted, unwalked -- only once
# EVERY pixel of the tile has stopped (T < 1e-4).  A stopped pixel's final_T is the transmittance BEFORE the stopping splat
# (>= 1e-4, < 1e-4 / (1 - 0.99)); the proxy below calls a pixel stopped when final_T < 1e-2 and n_contrib < the list length.
ft = np.asarray(d["final_T"], dtype=np.float32).reshape(H, W)
padT = np.zeros(pad.shape, np.float32)
padT[:H, :W] = ft
ft_t = padT.reshape(pad.shape[0] // 16, 16, pad.shape[1] // 16, 16).transpose(0, 2, 1, 3).reshape(-1, 256)[:T]
dies = (ft_t < 1e-2).all(axis=1) & nz & (need < length)


def generations(K0):
    """front K0 keys of every tile sorted and composited first; the rest only for tiles with a pixel still alive"""
    early = dies & (need <= K0)
    return round(float(np.where(early, np.minimum(length, K0), length).sum()) / max(1, int(length.sum())), 4)


out = {
    "workload": wl, "image": size, "P": int(cal[0].shape[0]), "N": int(length.sum()), "tiles": int(T), "tiles_nonempty": int(nz.sum()),
    "list_length": {"mean": round(float(length[nz].mean()), 1), **{f"p{k}": int(np.percentile(length[nz], k)) for k in q}},
    "needed_prefix": {"mean": round(float(need[nz].mean()), 1), **{f"p{k}": int(np.percentile(need[nz], k)) for k in q}},
    "needed_fraction_of_N": round(float(need.sum()) / max(1, int(length.sum())), 4),
    "needed_fraction_by_prefix_rounded_to": {str(b): round(float((np.minimum(length, (need + b - 1) // b * b)).sum()) / max(1, int(length.sum())), 4)
                                             for b in (128, 256, 512, 1024)},
    "tiles_whose_every_pixel_stops": int(dies.sum()), "their_share_of_N": round(float(length[dies].sum()) / max(1, int(length.sum())), 4),
    "realisable_sorted_fraction_two_generations_front_K0": {str(k): generations(k) for k in (512, 1024, 2048, 4096)},
    "realisable_sorted_fraction_tile_death_known_exactly": round(float(np.where(dies, need, length).sum()) / max(1, int(length.sum())), 4),
    "interactions_sum_n_contrib": int(nc.astype(np.int64).sum()),
    "oracle_seconds": round(time.time() - t0, 1),
}
print(json.dumps(out, indent=1))
