"""GPU-box diagnostic: c5-size parity by frame; locates differing pixels / tiles and checks the per-tile sorted lists
against the oracle's.  usage: python tools/diag_c5.py [workload]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (ROOT, os.path.join(ROOT, "gaussian-mesh-splatting_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import _util as U
from games_hip import synthetic as syn
from games_hip.model import HipGaussianMeshModel
from games_hip.render import PipelineParams, render
from oracle import gs_oracle, mesh_oracle
import diff_gaussian_rasterization as dgr
dgr.keep_buffers(True)        # this tool reads the scratch buffers of the last forward

wl = sys.argv[1] if len(sys.argv) > 1 else "c5_flame_like_1m"
scene = syn.mesh_scene(wl, state="trained")
size = scene.meta["image"]
cam = syn.orbit_camera(2, width=size, height=size)
with torch.no_grad():
    _, _, xyz, scaling, rot = mesh_oracle.mesh_to_gaussians(scene.vertices, scene.faces, scene._alpha, scene._scale)
    xa, sa, ra, oa, shs = mesh_oracle.activated(xyz, scaling, rot, scene._opacity, scene._features_dc, scene._features_rest)
okw = {k: v for k, v in U.settings_kwargs(cam, torch.ones(3)).items() if k not in ("prefiltered", "debug")}
o = gs_oracle.rasterize(means3D=xa, opacities=oa, shs=shs, scales=sa, rotations=ra, **okw)
det = o.state.details()
rng = det["ranges"]; depth = rng[:, 1] - rng[:, 0]
print("N", det["N"], "deepest", depth.max(), "tiles>8192:", int((depth > 8192).sum()), "tiles>1024:", int((depth > 1024).sum()), flush=True)
model = HipGaussianMeshModel.from_scene(scene, "cuda")
gx = (size + 15) // 16
al = lambda v: (v + 255) // 256 * 256
T = gx * gx
for frame in range(3):
    with torch.no_grad():
        model.update_alpha(); model.prepare_scaling_rot()
        pkg = render(cam.to("cuda"), model, PipelineParams(), torch.ones(3, device="cuda"))
    torch.cuda.synchronize()
    img = pkg["render"].cpu().numpy()
    diff = np.abs(img - o.color).max(0)
    amb = det["pix_ambig"].astype(bool)
    bad = (diff > 1e-4) & ~amb
    st = dgr.last_stats()
    print(f"frame {frame}: N_hip {st['num_rendered']} hint {st['capacity_hint']} bad clean pixels {int(bad.sum())} max {diff[~amb].max():.3e} radii mismatch {int((pkg['radii'].cpu().numpy() != o.radii).sum())}", flush=True)
    ys, xs = np.nonzero(bad)
    tiles = np.unique((ys // 16) * gx + xs // 16)
    print("  bad tiles:", len(tiles), "depths:", sorted(depth[tiles].tolist())[-10:], "min depth", depth[tiles].min() if len(tiles) else None)
    # n_contrib / final_T
    image_buf = dgr.raw_buffers()["image"]; HW = size * size
    fT = image_buf[:4 * HW].view(torch.float32).cpu().numpy().reshape(size, size)
    off = al(HW * 4)
    nc = image_buf[off:off + 4 * HW].view(torch.int32).cpu().numpy().reshape(size, size)
    print("  n_contrib mismatch (clean px):", int(((nc != det["n_contrib"]) & ~amb).sum()), " final_T maxdiff clean:", float(np.abs(fT - det["final_T"])[~amb].max()))
    # sorted lists
    binb = dgr.raw_buffers().get("binning")
    if binb is not None:
        toff_off = 2 * al(HW * 4) + 2 * al(T * 4)
        toff = image_buf[toff_off:toff_off + 4 * (T + 1)].view(torch.int32).cpu().numpy().astype(np.int64)
        N = int(st["num_rendered"])
        keys = binb[:8 * N].view(torch.int64).cpu().numpy()
        ids = (keys & 0xffffffff).astype(np.uint32)
        print("  tile_offset matches oracle ranges:", bool(np.array_equal(toff[:-1][depth > 0], rng[depth > 0, 0])), " last", toff[-1], N)
        nbad_t = 0; worst = []
        for t in np.nonzero(depth > 0)[0]:
            a = ids[toff[t]:toff[t + 1]]; b = det["point_list"][rng[t, 0]:rng[t, 1]]
            if not np.array_equal(a, b):
                nbad_t += 1
                if len(worst) < 8:
                    first = int(np.nonzero(a != b)[0][0]) if len(a) == len(b) else -1
                    worst.append((int(t), int(depth[t]), first, bool(np.array_equal(np.sort(a), np.sort(b)))))
        print("  tiles with a list != oracle:", nbad_t, "examples (tile, depth, first diff pos, same multiset):", worst, flush=True)
    geom = dgr.raw_buffers().get("geom")
    if geom is not None and frame == 0:
        P = scene.num_gaussians
        rec = geom[:48 * P].view(torch.float32).reshape(P, 12).cpu().numpy()
        vis = o.radii > 0
        hip_xy, hip_con = rec[vis][:, 0:2], rec[vis][:, [2, 3, 4, 5]]
        ora_xy, ora_con = det["xy"][vis].astype(np.float32), det["conic_op"][vis].astype(np.float32)
        for nm, a, b in (("xy", hip_xy, ora_xy), ("conic A", hip_con[:, 0], ora_con[:, 0]), ("conic B", hip_con[:, 1], ora_con[:, 1]),
                         ("conic C", hip_con[:, 2], ora_con[:, 2]), ("opacity", hip_con[:, 3], ora_con[:, 3]),
                         ("rgb", rec[vis][:, [6, 7, 8]], det["rgb"][vis].astype(np.float32))):
            neq = (a.view(np.uint32) != b.view(np.uint32))
            rel = np.abs(a - b) / (np.abs(b) + 1e-30)
            print(f"  record field {nm}: bit mismatches {int(neq.sum())} of {neq.size}, max rel {rel.max():.2e}")
    if len(ys):
        k = np.argmax(diff * bad)
        y, x = divmod(int(k), size)
        print("  worst pixel", (x, y), "diff", diff[y, x], "hip", img[:, y, x], "ora", o.color[:, y, x], "nc hip/ora", nc[y, x], det["n_contrib"][y, x], "T", fT[y, x], det["final_T"][y, x])
