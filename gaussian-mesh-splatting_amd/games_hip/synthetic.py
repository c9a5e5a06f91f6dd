"""Deterministic synthetic scenes for the rasterizer path.

The reference ships no dataset, mesh or FLAME model, so every benchmark / parity input is
generated here.  Camera construction restates the reference's conventions:
  utils/graphics_utils.py:38-49  getWorld2View2      (W2C, then transposed)
  utils/graphics_utils.py:51-71  getProjectionMatrix (z in [0,1], z_sign=+1)
  scene/cameras.py:48-57         world_view_transform = W2C^T, full_proj = W2C^T @ P^T,
                                 camera_center = inverse(W2C^T)[3,:3], znear=0.01, zfar=100
Splat parameters follow the reference initialisers:
  games/mesh_splatting/scene/dataset_readers.py:73-77   _alpha = rand(F, S, 3)
  games/mesh_splatting/scene/gaussian_mesh_model.py:60,70  _scale = 1, opacity = inverse_sigmoid(0.1)
Everything is produced on CPU (float32) and moved by the caller.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np
import torch

C0 = 0.28209479177387814  # utils/sh_utils.py:26


def RGB2SH(rgb):
    return (rgb - 0.5) / C0


def inverse_sigmoid(x):
    return math.log(x / (1 - x))


# --------------------------------------------------------------------------- meshes
def uv_sphere(n_lat: int, n_lon: int, radius: float = 1.0, bump: float = 0.05):
    """UV sphere with F = 2*n_lon*(n_lat-1) faces, radially displaced by
    bump*sin(7*theta)*cos(5*phi) (deterministic, no RNG)."""
    verts = [(0.0, 0.0, 1.0)]
    thetas = [math.pi * i / n_lat for i in range(1, n_lat)]
    for th in thetas:
        for j in range(n_lon):
            ph = 2 * math.pi * j / n_lon
            verts.append((math.sin(th) * math.cos(ph), math.sin(th) * math.sin(ph), math.cos(th)))
    verts.append((0.0, 0.0, -1.0))
    v = np.asarray(verts, dtype=np.float64)
    theta = np.arccos(np.clip(v[:, 2], -1, 1))
    phi = np.arctan2(v[:, 1], v[:, 0])
    r = radius * (1.0 + bump * np.sin(7 * theta) * np.cos(5 * phi))
    v = v * r[:, None]

    def ring(i, j):  # ring i in [0, n_lat-2]
        return 1 + i * n_lon + (j % n_lon)

    faces = []
    for j in range(n_lon):
        faces.append((0, ring(0, j), ring(0, j + 1)))
    for i in range(n_lat - 2):
        for j in range(n_lon):
            a, b, c, d = ring(i, j), ring(i, j + 1), ring(i + 1, j), ring(i + 1, j + 1)
            faces.append((a, c, d))
            faces.append((a, d, b))
    south = len(verts) - 1
    for j in range(n_lon):
        faces.append((south, ring(n_lat - 2, j + 1), ring(n_lat - 2, j)))
    return torch.tensor(v, dtype=torch.float32), torch.tensor(np.asarray(faces), dtype=torch.int64)


# --------------------------------------------------------------------------- cameras
@dataclass
class SynthCamera:
    """Duck-types the attributes `render()` reads from scene/cameras.py:Camera."""
    image_width: int
    image_height: int
    FoVx: float
    FoVy: float
    world_view_transform: torch.Tensor  # [4,4] = W2C^T
    projection_matrix: torch.Tensor     # [4,4] = P^T
    full_proj_transform: torch.Tensor   # [4,4]
    camera_center: torch.Tensor         # [3]
    znear: float = 0.01
    zfar: float = 100.0

    @property
    def tanfovx(self):
        return math.tan(self.FoVx * 0.5)

    @property
    def tanfovy(self):
        return math.tan(self.FoVy * 0.5)

    def to(self, device):
        return SynthCamera(self.image_width, self.image_height, self.FoVx, self.FoVy,
                           self.world_view_transform.to(device), self.projection_matrix.to(device),
                           self.full_proj_transform.to(device), self.camera_center.to(device),
                           self.znear, self.zfar)


def projection_matrix(znear, zfar, fovX, fovY):
    tanHalfFovY = math.tan(fovY / 2)
    tanHalfFovX = math.tan(fovX / 2)
    top, right = tanHalfFovY * znear, tanHalfFovX * znear
    bottom, left = -top, -right
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def look_at_camera(cam_pos, target=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0), width=800, height=800,
                   fovx=0.6911112070083618, znear=0.01, zfar=100.0) -> SynthCamera:
    c = np.asarray(cam_pos, dtype=np.float64)
    f = np.asarray(target, dtype=np.float64) - c
    f /= np.linalg.norm(f)
    r = np.cross(f, np.asarray(up, dtype=np.float64))
    r /= np.linalg.norm(r)
    d = np.cross(f, r)                       # camera "down" (COLMAP: x right, y down, z forward)
    Rot = np.stack([r, d, f])                # rows
    W2C = np.eye(4)
    W2C[:3, :3] = Rot
    W2C[:3, 3] = -Rot @ c
    focal = width / (2 * math.tan(fovx / 2))
    fovy = 2 * math.atan(height / (2 * focal))
    wvt = torch.tensor(np.float32(W2C)).transpose(0, 1).contiguous()
    proj = projection_matrix(znear, zfar, fovx, fovy).transpose(0, 1).contiguous()
    full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).contiguous()
    center = wvt.inverse()[3, :3].contiguous()
    return SynthCamera(width, height, fovx, fovy, wvt, proj, full, center, znear, zfar)


def orbit_camera(k: int, n_views: int = 8, radius: float = 4.0311, elevation_deg: float = 30.0,
                 width: int = 800, height: int = 800, fovx: float = 0.6911112070083618) -> SynthCamera:
    az = 2 * math.pi * k / n_views
    el = math.radians(elevation_deg)
    pos = (radius * math.cos(el) * math.cos(az), radius * math.cos(el) * math.sin(az), radius * math.sin(el))
    return look_at_camera(pos, width=width, height=height, fovx=fovx)


# --------------------------------------------------------------------------- scenes
@dataclass
class MeshScene:
    """Learnable tensors of a GaussianMeshModel-like scene (mesh-bound Gaussians)."""
    vertices: torch.Tensor        # [V,3]
    faces: torch.Tensor           # [F,3] int64
    _alpha: torch.Tensor          # [F,S,3]
    _scale: torch.Tensor          # [P,1]
    _opacity: torch.Tensor        # [P,1]  (pre-sigmoid)
    _features_dc: torch.Tensor    # [P,1,3]
    _features_rest: torch.Tensor  # [P,15,3]
    active_sh_degree: int = 3
    alpha_mode: str = "relu"
    meta: Dict = field(default_factory=dict)

    @property
    def num_gaussians(self):
        return self._alpha.shape[0] * self._alpha.shape[1]

    def to(self, device):
        return MeshScene(self.vertices.to(device), self.faces.to(device), self._alpha.to(device),
                         self._scale.to(device), self._opacity.to(device), self._features_dc.to(device),
                         self._features_rest.to(device), self.active_sh_degree, self.alpha_mode, dict(self.meta))


MESH_CONFIGS = {
    # name: (n_lat, n_lon, splats_per_face, image size)
    "c2_hotdog_like": (224, 224, 3, 800),     # F = 99 904, P = 299 712
    "c5_flame_like_500k": (59, 86, 50, 1024),  # F = 9 976, P = 498 800
    "c5_flame_like_1m": (59, 86, 100, 1024),   # F = 9 976, P = 997 600
    "small": (24, 24, 3, 128),                 # F = 1 104, P = 3 312
    "tiny": (8, 10, 2, 64),                    # F = 140, P = 280
}


def mesh_scene(name: str = "c2_hotdog_like", state: str = "trained", seed: int = 0,
               n_lat: Optional[int] = None, n_lon: Optional[int] = None, splats: Optional[int] = None) -> MeshScene:
    """`state="init"`: exactly the reference's initialisation (opacity 0.1, _scale 1, zero higher SH).
    `state="trained"`: trained-like statistics."""
    cfg = MESH_CONFIGS[name]
    n_lat = n_lat or cfg[0]
    n_lon = n_lon or cfg[1]
    S = splats or cfg[2]
    vertices, faces = uv_sphere(n_lat, n_lon)
    F = faces.shape[0]
    P = F * S
    g = torch.Generator().manual_seed(seed)
    _alpha = torch.rand(F, S, 3, generator=g)
    if state == "init":
        _scale = torch.ones(P, 1)
        _opacity = torch.full((P, 1), inverse_sigmoid(0.1))
        f_dc = RGB2SH(torch.rand(P, 1, 3, generator=g))
        f_rest = torch.zeros(P, 15, 3)
    elif state == "trained":
        _scale = torch.exp(0.3 * torch.randn(P, 1, generator=g))
        _opacity = 1.0 + 2.0 * torch.randn(P, 1, generator=g)
        f_dc = RGB2SH(torch.rand(P, 1, 3, generator=g))
        f_rest = 0.05 * torch.randn(P, 15, 3, generator=g)
    else:
        raise ValueError(state)
    return MeshScene(vertices, faces, _alpha, _scale, _opacity, f_dc, f_rest, 3, "relu",
                     {"name": name, "state": state, "F": F, "S": S, "P": P, "image": cfg[3]})


MULTI_MESH_CONFIGS = {
    # name: ([(n_lat, n_lon, splats_per_face, radius, (cx, cy, cz))], image size).  BASELINE config 4 ("gs_multi_mesh
    # ficus, ~300k Gaussians, 800x800"): three meshes with different splat counts per face, 299 472 Gaussians
    "c4_ficus_like": ([(130, 130, 3, 0.62, (0.0, 0.0, 0.45)), (100, 100, 5, 0.5, (0.45, -0.3, -0.4)),
                       (160, 157, 2, 0.55, (-0.45, 0.3, -0.35))], 800),
    "multi_tiny": ([(6, 8, 2, 0.6, (0.0, 0.0, 0.4)), (5, 6, 3, 0.5, (0.4, -0.2, -0.4))], 64),
}


def multi_mesh_scenes(name: str = "c4_ficus_like", state: str = "trained", seed: int = 0):
    """List of MeshScene (one per mesh) for the multi-mesh model: each mesh keeps its own faces / splats-per-face
    (games/multi_mesh_splatting/scene/gaussian_multi_mesh_model.py:48-97 keeps them as python lists)."""
    parts, size = MULTI_MESH_CONFIGS[name]
    scenes = []
    for k, (n_lat, n_lon, S, radius, centre) in enumerate(parts):
        sc = mesh_scene("c2_hotdog_like", state=state, seed=seed + k, n_lat=n_lat, n_lon=n_lon, splats=S)
        sc.vertices = sc.vertices * radius + torch.tensor(centre, dtype=torch.float32)
        sc.meta.update({"name": f"{name}[{k}]", "image": size})
        scenes.append(sc)
    return scenes


@dataclass
class FreeScene:
    """Free (non mesh-bound) Gaussians in the rasterizer's own input terms."""
    means3D: torch.Tensor    # [P,3]
    scales: torch.Tensor     # [P,3]   (activated)
    rotations: torch.Tensor  # [P,4]   (normalised, w first)
    opacities: torch.Tensor  # [P,1]   (activated)
    shs: torch.Tensor        # [P,16,3]
    sh_degree: int = 3

    def to(self, device):
        return FreeScene(*(t.to(device) for t in (self.means3D, self.scales, self.rotations, self.opacities, self.shs)),
                         self.sh_degree)


def flat_scene(P: int = 10_000, seed: int = 0) -> FreeScene:
    """BASELINE config 1: gs_flat-like random Gaussians (first scale axis pinned to 1e-8,
    games/flat_splatting/scene/flat_gaussian_model.py:32-35; xyz ~ U(-1.3,1.3)^3, scene/dataset_readers.py:240)."""
    g = torch.Generator().manual_seed(seed)
    xyz = (torch.rand(P, 3, generator=g) * 2.6) - 1.3
    s = torch.exp(torch.log(0.005 + 0.045 * torch.rand(P, 2, generator=g)))
    scales = torch.cat([torch.full((P, 1), 1e-8), s], dim=1)
    q = torch.nn.functional.normalize(torch.randn(P, 4, generator=g))
    op = torch.full((P, 1), 0.1)
    f_dc = RGB2SH(torch.rand(P, 1, 3, generator=g))
    f_rest = 0.05 * torch.randn(P, 15, 3, generator=g)
    return FreeScene(xyz, scales, q, op, torch.cat([f_dc, f_rest], dim=1).contiguous(), 3)


def random_scene(P: int, seed: int = 0, extent: float = 1.0, scale_lo: float = 0.01, scale_hi: float = 0.15,
                 opacity_lo: float = 0.05, opacity_hi: float = 1.0, sh_rest_std: float = 0.2) -> FreeScene:
    """Generic anisotropic random Gaussians for parity tests (large footprints, mixed opacity)."""
    g = torch.Generator().manual_seed(seed)
    xyz = (torch.rand(P, 3, generator=g) * 2 - 1) * extent
    scales = scale_lo * (scale_hi / scale_lo) ** torch.rand(P, 3, generator=g)
    q = torch.nn.functional.normalize(torch.randn(P, 4, generator=g))
    op = opacity_lo + (opacity_hi - opacity_lo) * torch.rand(P, 1, generator=g)
    f_dc = RGB2SH(torch.rand(P, 1, 3, generator=g))
    f_rest = sh_rest_std * torch.randn(P, 15, 3, generator=g)
    return FreeScene(xyz, scales, q, op, torch.cat([f_dc, f_rest], dim=1).contiguous(), 3)


def upstream_grad(image: torch.Tensor) -> torch.Tensor:
    """Deterministic dense dL/dcolor used for fwd+bwd timing and parity: (image - 0.5)/(3HW)."""
    return (image - 0.5) / image.numel()
