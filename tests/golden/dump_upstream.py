"""Golden vectors from the UPSTREAM CUDA rasterizer (round-5 review, item 7).

The rasterizer the reference calls lives in the un-vendored submodule `submodules/diff-gaussian-rasterization`
(graphdeco-inria/diff-gaussian-rasterization; empty in the reference tree, no CUDA toolchain in this image), so nothing under
tests/golden/ pins K1-K9 against the CUDA binary: DESIGN.md says "parity unpinned".  This script is what lifts that cap.  Run it ONCE on
any machine that has the upstream module built (a CUDA box with the reference's environment):

    python tests/golden/dump_upstream.py [--module-dir /path/that/contains/diff_gaussian_rasterization] [--out tests/golden]

It renders the suite's own scenes -- the six parametrised cases of tests/test_gpu_raster.py, the precomputed-colour / cov3D case, BASELINE
config 1 (gs_flat, 10 000 Gaussians, 256x256) and a 1 000-Gaussian slice of the config-2 hotdog-like model at 800x800 -- through
`GaussianRasterizer` forward AND backward (the upstream gradient of SURVEY 8(d), dL/dcolor = (image - 0.5) / (3HW) x 1000, plus a constant
dL/dinvdepth where the module returns an inverse-depth map) and writes one `upstream_<case>.npz` per case: every input, the 13 settings
fields, the three outputs, the gradient that was fed in and the eight gradients that came back, plus where they came from (module file,
torch / CUDA versions, device name).  Commit the files next to this script.

Consumers (both skip, and say "parity unpinned", while no file exists):
  tests/test_oracle_upstream_golden.py  (CPU suite)  the C oracle against the dump -- this is what PINS the oracle;
  tests/test_gpu_upstream_golden.py     (GPU suite)  the HIP kernels against the dump, tolerances of BASELINE.json's north_star
                                                     (1e-4 abs on RGB, 1e-3 rel on gradients) under the suite's discontinuity rule.
`--self` renders with whatever `diff_gaussian_rasterization` is first on sys.path INCLUDING this repository's drop-in: the GPU test uses
it into a temporary directory to prove the plumbing end to end (a dump of ourselves pins nothing and is never committed).

Only `games_hip/synthetic.py` (by file path: scene generators, pure torch) and `oracle/mesh_oracle.py` (the mesh -> Gaussian restatement, pure
torch, for the hotdog slice) of this repository are loaded; the rasterizer module is NOT taken from this repository unless --self is given."""
import argparse
import importlib
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
PKG = os.path.join(ROOT, "gaussian-mesh-splatting_amd")
SETTINGS = ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix", "sh_degree", "campos",
            "prefiltered", "debug", "antialiasing")
GRADS = ("means3D", "means2D", "shs", "colors_precomp", "opacities", "scales", "rotations", "cov3D_precomp")


def _synthetic():
    spec = importlib.util.spec_from_file_location("_gms_synthetic", os.path.join(PKG, "games_hip", "synthetic.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["_gms_synthetic"] = mod          # (dataclasses look their module up by name)
    spec.loader.exec_module(mod)
    return mod


def settings_kwargs(cam, bg, sh_degree=3, antialiasing=False, scale_modifier=1.0):
    return dict(image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=float(cam.tanfovx), tanfovy=float(cam.tanfovy),
                bg=bg, scale_modifier=float(scale_modifier), viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
                sh_degree=int(sh_degree), campos=cam.camera_center, prefiltered=False, debug=False, antialiasing=bool(antialiasing))


def cov3d_python(scales, mod, rotations):
    """scene/gaussian_model.py:27-31 + utils/general_utils.py:144-190 (the reference's python cov3D path), on any device."""
    q = rotations / rotations.norm(dim=1, keepdim=True)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    L = R * (mod * scales)[:, None, :]
    S = L @ L.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1)


def cases(syn):
    """name -> (inputs, settings).  The first six are tests/test_gpu_raster.py::test_forward_backward_parity's, seed for seed."""
    def free(sc):
        return dict(means3D=sc.means3D, opacities=sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    bg = torch.tensor([0.2, 0.4, 0.6])
    out = {}
    out["random"] = (free(syn.random_scene(4000, seed=1, scale_lo=0.01, scale_hi=0.1)), settings_kwargs(syn.orbit_camera(1, width=160, height=128, radius=3.0), bg))
    out["aa_deg2"] = (free(syn.random_scene(3000, seed=2, scale_lo=0.005, scale_hi=0.08)),
                      settings_kwargs(syn.orbit_camera(2, width=200, height=120, radius=3.0), bg, antialiasing=True, sh_degree=2))
    out["deg0"] = (free(syn.random_scene(2000, seed=3)), settings_kwargs(syn.orbit_camera(3, width=96, height=96, radius=3.5), bg, sh_degree=0))
    out["flat10k"] = (free(syn.flat_scene(10000)), settings_kwargs(syn.orbit_camera(0, width=256, height=256), torch.ones(3)))          # BASELINE config 1
    out["odd_size"] = (free(syn.random_scene(2500, seed=5, scale_lo=0.02, scale_hi=0.2)), settings_kwargs(syn.orbit_camera(5, width=131, height=77, radius=3.0), bg))
    out["scale_mod"] = (free(syn.random_scene(2000, seed=6)), settings_kwargs(syn.orbit_camera(6, width=128, height=128, radius=3.0), bg, scale_modifier=1.7))
    sc = syn.random_scene(3000, seed=7, scale_lo=0.01, scale_hi=0.12)
    out["precomp"] = (dict(means3D=sc.means3D, opacities=sc.opacities, colors_precomp=torch.rand(3000, 3, generator=torch.Generator().manual_seed(0)),
                           cov3D_precomp=cov3d_python(sc.scales, 1.0, sc.rotations)),
                      settings_kwargs(syn.orbit_camera(4, width=144, height=112, radius=3.0), torch.zeros(3), sh_degree=0))
    return out


def hotdog_slice(syn):
    """BASELINE config 3's "gradcheck vs reference on a 1k-Gaussian slice": every 300th Gaussian of the config-2 model's rasterizer inputs
    -- derived by oracle/mesh_oracle.py, the restatement of games/mesh_splatting/scene/gaussian_mesh_model.py:86-169 that reproduces the
    reference's own classes bit for bit on tests/golden/k0_*.npz -- at 800x800."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import mesh_oracle
    sc = syn.mesh_scene("c2_hotdog_like", state="trained")
    _, _, xyz, scaling, rotation = mesh_oracle.mesh_to_gaussians(sc.vertices, sc.faces.long(), sc._alpha, sc._scale, sc.alpha_mode)
    xyz, scales, rots, opac, shs = mesh_oracle.activated(xyz, scaling, rotation, sc._opacity, sc._features_dc, sc._features_rest)
    keep = torch.arange(0, xyz.shape[0], 300)
    inputs = dict(means3D=xyz[keep], opacities=opac[keep], shs=shs[keep], scales=scales[keep], rotations=rots[keep])
    return inputs, settings_kwargs(syn.orbit_camera(0, width=800, height=800), torch.ones(3))


def run_case(dgr, inputs, kw, device):
    """One forward + backward through `dgr.GaussianRasterizer`; returns a dict of numpy arrays (what the npz holds)."""
    t = {k: v.to(device).float().detach().clone().requires_grad_(True) for k, v in inputs.items()}
    kwd = {k: (v.to(device).float() if torch.is_tensor(v) else v) for k, v in kw.items()}
    means2D = torch.zeros_like(t["means3D"], requires_grad=True)
    res = dgr.GaussianRasterizer(raster_settings=dgr.GaussianRasterizationSettings(**kwd))(
        means3D=t["means3D"], means2D=means2D, opacities=t["opacities"], shs=t.get("shs"), colors_precomp=t.get("colors_precomp"),
        scales=t.get("scales"), rotations=t.get("rotations"), cov3D_precomp=t.get("cov3D_precomp"))
    color, radii = res[0], res[1]
    invd = res[2] if len(res) > 2 else None
    H, W = color.shape[1], color.shape[2]
    gc = ((color.detach() - 0.5) / (3.0 * H * W)) * 1000.0          # SURVEY 8(d)'s upstream gradient (x 1000, as the suite)
    loss = (color * gc).sum()
    gd = None
    if invd is not None:
        gd = torch.full_like(invd, 1e-3)
        loss = loss + (invd * gd).sum()
    loss.backward()
    out = {"in_" + k: v.detach().cpu().numpy() for k, v in inputs.items()}
    for k in SETTINGS:
        v = kw[k]
        out["set_" + k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    out["out_color"] = color.detach().cpu().numpy()
    out["out_radii"] = radii.detach().cpu().numpy()
    if invd is not None:
        out["out_invdepth"] = invd.detach().cpu().numpy()
        out["grad_invdepth"] = gd.cpu().numpy()
    out["grad_color"] = gc.cpu().numpy()
    for k in GRADS:
        src = means2D if k == "means2D" else t.get(k)
        if src is not None and src.grad is not None:
            out["dL_" + k] = src.grad.detach().cpu().numpy()
    return out


def load(path):
    """npz -> (inputs: dict of CPU tensors, settings kwargs, outputs dict, grads dict, provenance str)."""
    z = np.load(path, allow_pickle=False)
    inputs = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    kw = {}
    for k in SETTINGS:
        v = z["set_" + k]
        kw[k] = torch.from_numpy(v) if v.ndim > 0 else (bool(v) if v.dtype == np.bool_ else (int(v) if np.issubdtype(v.dtype, np.integer) else float(v)))
    outs = {k[4:]: z[k] for k in z.files if k.startswith("out_")}
    outs["grad_color"] = z["grad_color"]
    outs["grad_invdepth"] = z["grad_invdepth"] if "grad_invdepth" in z.files else None
    grads = {k[3:]: z[k] for k in z.files if k.startswith("dL_")}
    return inputs, kw, outs, grads, str(z["provenance"])


def dump_all(out_dir, self_module=False, module_dir=None, device="cuda", only=None):
    if module_dir:
        sys.path.insert(0, module_dir)
    elif self_module and PKG not in sys.path:
        sys.path.insert(0, PKG)
    dgr = importlib.import_module("diff_gaussian_rasterization")
    mfile = os.path.abspath(getattr(dgr, "__file__", "?"))
    if not self_module and mfile.startswith(PKG):
        raise SystemExit("diff_gaussian_rasterization resolves to THIS repository's drop-in (" + mfile + "): a dump of ourselves pins nothing. "
                         "Install / point --module-dir at the upstream CUDA build, or pass --self for the plumbing test.")
    syn = _synthetic()
    todo = dict(cases(syn))
    todo["hotdog_slice_1k"] = hotdog_slice(syn)
    prov = (f"module {mfile}; torch {torch.__version__}; hip {getattr(torch.version, 'hip', None)} cuda {getattr(torch.version, 'cuda', None)}; "
            f"device {torch.cuda.get_device_name(0) if torch.cuda.is_available() else 'cpu'}; self={bool(self_module)}")
    os.makedirs(out_dir, exist_ok=True)
    written = []
    for name, (inputs, kw) in todo.items():
        if only and name not in only:
            continue
        arrays = run_case(dgr, inputs, kw, device)
        arrays["provenance"] = np.asarray(prov)
        path = os.path.join(out_dir, f"upstream_{name}.npz")
        np.savez_compressed(path, **arrays)
        written.append(path)
        print(f"{name:18s} P={inputs['means3D'].shape[0]:6d} {kw['image_width']}x{kw['image_height']}  -> {path} ({os.path.getsize(path) / 1e6:.2f} MB)")
    return written


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=HERE)
    ap.add_argument("--module-dir", default=None, help="directory that contains the upstream diff_gaussian_rasterization package")
    ap.add_argument("--self", dest="self_module", action="store_true", help="plumbing test: render with this repository's own drop-in")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--only", nargs="*", default=None)
    a = ap.parse_args()
    dump_all(a.out, a.self_module, a.module_dir, a.device, a.only)
