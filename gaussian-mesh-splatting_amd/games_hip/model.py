"""Mesh-bound Gaussian model on the fused HIP op: the attribute / method surface of the reference's
GaussianMeshModel that the render path touches, without the reference's dataset / PLY / optimizer code.

Mirrors (same names, same meaning):
  scene/gaussian_model.py:95-115       get_scaling / get_rotation / get_xyz / get_features / get_opacity
  games/mesh_splatting/scene/gaussian_mesh_model.py:153-169  update_alpha()
  games/mesh_splatting/scene/gaussian_mesh_model.py:103-151  prepare_scaling_rot()
  games/mesh_splatting/scene/gaussian_mesh_model.py:171-183  training_setup() parameter groups
`HipMeshMixin` can also be mixed into the reference's own classes (see games_hip.install).
"""
from __future__ import annotations

import torch
from torch import nn

from .mesh_op import mesh_to_gaussians, triangles_to_gaussians


class HipMeshMixin:
    """Overrides only update_alpha / prepare_scaling_rot.  Host class must provide: vertices, faces,
    _alpha [F,S,3], _scale [P,1]; optional `alpha_mode` ("relu" default, "softmax" for FLAME)."""

    alpha_mode = "relu"

    def update_alpha(self):
        opa = getattr(self, "_opacity", None)
        fuse_opacity = torch.is_tensor(opa) and opa.is_cuda and opa.numel() == self._scale.numel()
        out = mesh_to_gaussians(self.vertices, self.faces, self._alpha, self._scale, self.alpha_mode, fused_activations=True,
                                _opacity=opa if fuse_opacity else None)
        alpha, xyz, scaling, rotation, scaling_act, rotation_unit = out[:6]
        # get_opacity (scene/gaussian_model.py:113-115) from the same kernel; valid while _opacity is unchanged
        self._hip_opacity = (opa, opa._version, out[6]) if fuse_opacity else None
        self.alpha = alpha
        self._xyz = xyz
        # `triangles` is only read by save_ply and the animated renderers: gathered on first access
        self.__dict__.pop("_hip_tri", None)
        self._hip_tri_external = None
        self._hip_cached = (scaling, rotation, scaling_act, rotation_unit)

    @property
    def triangles(self):
        ext = self.__dict__.get("_hip_tri_external")
        if ext is not None:
            return ext
        tri = self.__dict__.get("_hip_tri")
        if tri is None and getattr(self, "vertices", None) is not None and getattr(self, "faces", None) is not None \
                and torch.is_tensor(self.faces) and self.faces.numel():
            with torch.no_grad():
                tri = self.vertices[self.faces]
            self.__dict__["_hip_tri"] = tri
        return tri

    @triangles.setter
    def triangles(self, value):
        # a renderer / loader replaced pc.triangles (renderer/gaussian_animated_renderer/__init__.py:72)
        self.__dict__["_hip_tri_external"] = value

    def prepare_scaling_rot(self, *unused):
        tri = self.__dict__.get("_hip_tri_external")
        cached = getattr(self, "_hip_cached", None)
        if tri is not None:
            # a renderer replaced pc.triangles (renderer/gaussian_animated_renderer/__init__.py:72-73):
            # derive scale / rotation from those triangles
            _, _, scaling, rotation, scaling_act, rotation_unit = triangles_to_gaussians(
                tri, self._alpha, self._scale, self.alpha_mode, fused_activations=True)
        elif cached is not None:
            scaling, rotation, scaling_act, rotation_unit = cached
        else:
            _, _, scaling, rotation, scaling_act, rotation_unit = mesh_to_gaussians(
                self.vertices, self.faces, self._alpha, self._scale, self.alpha_mode, fused_activations=True)
        self._scaling = scaling
        self._rotation = rotation
        self._hip_activated = (scaling, rotation, scaling_act, rotation_unit)

    # Property getters (scene/gaussian_model.py:95-101): exp / normalize were computed by the fused
    # kernel; fall back to the reference's formula if somebody replaced _scaling / _rotation.
    @property
    def get_scaling(self):
        act = getattr(self, "_hip_activated", None)
        if act is not None and act[0] is self._scaling:
            return act[2]
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        act = getattr(self, "_hip_activated", None)
        if act is not None and act[1] is self._rotation:
            return act[3]
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_opacity(self):
        cached = getattr(self, "_hip_opacity", None)
        if cached is not None and cached[0] is self._opacity and cached[1] == self._opacity._version:
            return cached[2]
        return torch.sigmoid(self._opacity)

    @property
    def get_features(self):
        # scene/gaussian_model.py:107-111 concatenates 57.6 MB per iteration; the rasterizer reads and
        # differentiates _features_dc / _features_rest in place instead (SplitSH behaves as the concatenation
        # for any other consumer)
        from diff_gaussian_rasterization import SplitSH
        return SplitSH(self._features_dc, self._features_rest)


class HipMultiMeshMixin:
    """Drop-in for GaussianMultiMeshModel.update_alpha / _calc_xyz / prepare_scaling_rot
    (games/multi_mesh_splatting/scene/gaussian_multi_mesh_model.py:99-199): the per-mesh python loop and the
    torch.cat of its results become ONE launch over the concatenated meshes (faces re-indexed by the vertex
    offsets, splat ranges as CSR because every mesh may carry a different number of splats per face).
    Host class provides the reference's list attributes: vertices[i], faces[i], _alpha[i] [F_i,S_i,3], _scale[i] [P_i,1]."""

    def _hip_topology(self):
        key = tuple((int(f.shape[0]), int(a.shape[1]), int(v.shape[0])) for f, a, v in zip(self.faces, self._alpha, self.vertices))
        cached = self.__dict__.get("_hip_topo")
        if cached is not None and cached[0] == key:
            return cached[1:]
        device = self.vertices[0].device
        faces, counts = [], []
        voff = 0
        for f, a, v in zip(self.faces, self._alpha, self.vertices):
            faces.append(f.to(device).long() + voff)
            counts.append(torch.full((int(f.shape[0]),), int(a.shape[1]), dtype=torch.int64))
            voff += int(v.shape[0])
        counts = torch.cat(counts)
        fso = torch.cat([torch.zeros(1, dtype=torch.int64), counts.cumsum(0)]).to(torch.int32)
        sf = torch.repeat_interleave(torch.arange(counts.numel(), dtype=torch.int32), counts)
        topo = (torch.cat(faces).contiguous(), fso.to(device), sf.to(device))
        self.__dict__["_hip_topo"] = (key,) + topo
        return topo

    def update_alpha(self):
        faces, fso, sf = self._hip_topology()
        V = torch.cat(list(self.vertices))
        A = torch.cat([a.reshape(-1, 3) for a in self._alpha])
        Sc = torch.cat(list(self._scale))
        alpha, xyz, scaling, rotation, scaling_act, rotation_unit = mesh_to_gaussians(
            V, faces, A, Sc, "relu", face_splat_offset=fso, splat_face=sf, fused_activations=True)
        sizes = [a.shape[0] * a.shape[1] for a in self._alpha]
        self.alpha = [x.reshape(a.shape) for x, a in zip(torch.split(alpha, sizes), self._alpha)]
        self._xyz = xyz
        self._hip_cached = (scaling, rotation, scaling_act, rotation_unit)

    def prepare_scaling_rot(self, *unused):
        if self.__dict__.get("_hip_cached") is None:
            self.update_alpha()
        scaling, rotation, scaling_act, rotation_unit = self._hip_cached
        self._scaling, self._rotation = scaling, rotation
        self._hip_activated = (scaling, rotation, scaling_act, rotation_unit)

    get_scaling = HipMeshMixin.get_scaling
    get_rotation = HipMeshMixin.get_rotation


class HipGaussianMeshModel(HipMeshMixin):
    """Stand-alone model (no dependency on the reference tree) used by bench.py / tests."""

    def __init__(self, sh_degree: int = 3):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        self.optimizer = None

    @classmethod
    def from_scene(cls, scene, device="cuda"):
        m = cls(3)
        m.active_sh_degree = scene.active_sh_degree
        m.alpha_mode = scene.alpha_mode
        m.vertices = nn.Parameter(scene.vertices.to(device).float().contiguous())
        m.faces = scene.faces.to(device)
        m._alpha = nn.Parameter(scene._alpha.to(device).float().contiguous())
        m._scale = nn.Parameter(scene._scale.to(device).float().contiguous())
        m._opacity = nn.Parameter(scene._opacity.to(device).float().contiguous())
        m._features_dc = nn.Parameter(scene._features_dc.to(device).float().contiguous())
        m._features_rest = nn.Parameter(scene._features_rest.to(device).float().contiguous())
        m.update_alpha()
        m.prepare_scaling_rot()
        return m

    def parameters(self):
        return [self.vertices, self._alpha, self._features_dc, self._features_rest, self._opacity, self._scale]

    def training_setup(self, vertices_lr=0.0, alpha_lr=0.001, feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005,
                       fused=True):
        groups = [
            {"params": [self.vertices], "lr": vertices_lr, "name": "vertices"},
            {"params": [self._alpha], "lr": alpha_lr, "name": "alpha"},
            {"params": [self._features_dc], "lr": feature_lr, "name": "f_dc"},
            {"params": [self._features_rest], "lr": feature_lr / 20.0, "name": "f_rest"},
            {"params": [self._opacity], "lr": opacity_lr, "name": "opacity"},
            {"params": [self._scale], "lr": scaling_lr, "name": "scaling"},
        ]
        if fused:       # one HIP launch per step (csrc/adam.hip); same state layout as torch.optim.Adam
            from .optim import FusedAdam
            self.optimizer = FusedAdam(groups, lr=0.0, eps=1e-15)
        else:
            self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)

    # ---- checkpoints: the reference's on-disk format (scene/gaussian_model.py:177-268 point_cloud.ply +
    # games/mesh_splatting/scene/gaussian_mesh_model.py:189-222 model_params.pt), through the plyfile stand-in
    def construct_list_of_attributes(self):
        names = ["x", "y", "z", "nx", "ny", "nz"]
        names += [f"f_dc_{i}" for i in range(self._features_dc.shape[1] * self._features_dc.shape[2])]
        names += [f"f_rest_{i}" for i in range(self._features_rest.shape[1] * self._features_rest.shape[2])]
        names += ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]
        return names

    def save_ply(self, path):
        import os
        import numpy as np
        from plyfile import PlyData, PlyElement
        self.update_alpha()
        self.prepare_scaling_rot()
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        xyz = self._xyz.detach().cpu().numpy()
        cols = [xyz, np.zeros_like(xyz),
                self._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy(),
                self._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy(),
                self._opacity.detach().cpu().numpy(), self._scaling.detach().cpu().numpy(), self._rotation.detach().cpu().numpy()]
        elements = np.empty(xyz.shape[0], dtype=[(a, "f4") for a in self.construct_list_of_attributes()])
        elements[:] = list(map(tuple, np.concatenate(cols, axis=1)))
        PlyData([PlyElement.describe(elements, "vertex")]).write(path)
        torch.save({"_alpha": self._alpha, "_scale": self._scale, "point_cloud": None, "triangles": self.triangles,
                    "vertices": self.vertices, "faces": self.faces}, path.replace("point_cloud.ply", "model_params.pt"))

    def load_ply(self, path, device="cuda"):
        import numpy as np
        from plyfile import PlyData
        el = PlyData.read(path).elements[0]
        col = lambda n: np.asarray(el[n], dtype=np.float32)
        P = el.count
        f_dc = np.stack([col(f"f_dc_{i}") for i in range(3)], axis=1).reshape(P, 3, 1)
        rest_names = sorted([p.name for p in el.properties if p.name.startswith("f_rest_")], key=lambda x: int(x.split("_")[-1]))
        assert len(rest_names) == 3 * (self.max_sh_degree + 1) ** 2 - 3
        f_rest = np.stack([col(n) for n in rest_names], axis=1).reshape(P, 3, (self.max_sh_degree + 1) ** 2 - 1)
        par = lambda a: nn.Parameter(torch.tensor(a, dtype=torch.float, device=device).contiguous().requires_grad_(True))
        self._features_dc = par(np.transpose(f_dc, (0, 2, 1)))
        self._features_rest = par(np.transpose(f_rest, (0, 2, 1)))
        self._opacity = par(col("opacity")[:, None])
        params = torch.load(path.replace("point_cloud.ply", "model_params.pt"), map_location=device, weights_only=False)
        self.vertices = nn.Parameter(params["vertices"].detach().to(device))
        self.faces = params["faces"].to(device)
        self._alpha = nn.Parameter(params["_alpha"].detach().to(device))
        self._scale = nn.Parameter(params["_scale"].detach().to(device))
        self.active_sh_degree = self.max_sh_degree
        self.update_alpha()
        self.prepare_scaling_rot()

    @property
    def get_xyz(self):
        return self._xyz



def install(games_module=None):
    """Swap the fused op into the reference's registry (games/__init__.py:35-51): the mesh model keeps
    its class, dataset reader, optimizer groups and PLY I/O; only update_alpha / prepare_scaling_rot
    are overridden.  Call after `import games` in an environment that has the reference on sys.path."""
    if games_module is None:
        import games as games_module  # type: ignore
    base = games_module.gaussianModel["gs_mesh"]
    cls = type("Hip" + base.__name__, (HipMeshMixin, base), {"alpha_mode": "relu"})
    games_module.gaussianModel["gs_mesh"] = cls
    return {"gs_mesh": cls}
