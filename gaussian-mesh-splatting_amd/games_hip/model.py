"""Mesh-bound Gaussian models on the fused HIP op.

Three mixins override, with identical results, the K0 methods of the reference's three mesh-bound model classes and
nothing else (dataset readers, optimizer groups, PLY I/O stay the host class's):

  HipMeshMixin       GaussianMeshModel        games/mesh_splatting/scene/gaussian_mesh_model.py:86-169
  HipMultiMeshMixin  GaussianMultiMeshModel   games/multi_mesh_splatting/scene/gaussian_multi_mesh_model.py:99-199
  HipFlameMixin      GaussianFlameModel       games/flame_splatting/scene/gaussian_flame_model.py:123-207

`install()` puts them into both registries of games/__init__.py:35-51 (`gaussianModel` used by train.py,
`gaussianModelRender` used by scripts/render.py:22,41).  The stand-alone classes at the bottom
(`HipGaussianMeshModel`, `HipGaussianMultiMeshModel`, `HipGaussianFlameModel`) carry the same mixins on minimal hosts
for bench.py / the GPU tests, where the reference tree is absent.

Property getters fused into the op (scene/gaussian_model.py:95-115): get_scaling / get_rotation / get_opacity.
Caches are validated against the identity AND autograd version of every tensor they were derived from, so editing
`vertices`, `_scale`, `_alpha` or `_opacity` (an optimizer step, a checkpoint load) can never serve stale values.
"""
from __future__ import annotations

import torch
from torch import nn

from .mesh_op import mesh_to_gaussians, triangles_to_gaussians


class _Stamp:
    """The tensors a cached value was derived from, with their in-place versions.  Holds the tensors themselves
    (not their ids: an id can be reused once a tensor is freed), compared by identity + version."""

    __slots__ = ("items",)

    def __init__(self, *tensors):
        self.items = tuple((t, t._version if torch.is_tensor(t) else -1) for t in tensors)

    def __eq__(self, other):
        return (isinstance(other, _Stamp) and len(self.items) == len(other.items)
                and all(a is b and va == vb for (a, va), (b, vb) in zip(self.items, other.items)))

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None


def _stamp(*tensors):
    return _Stamp(*tensors)


class _HipGetters:
    """get_scaling / get_rotation / get_opacity / get_features on the fused outputs; each falls back to the
    reference's formula when `_scaling` / `_rotation` / `_opacity` is not the tensor the fused value came from."""

    # ---- deferred K0 (HipMeshMixin.hip_defer_k0): the derived attributes are materialised when somebody asks for them
    def _hip_materialize(self):
        pass

    def _hip_frame_value(self, k):
        """While K0 is deferred and nothing is being differentiated (an evaluation pass between two training steps), the getters
        serve what the last training frame derived -- if the inputs have not changed since -- instead of launching K0."""
        fr = self.__dict__.get("_hip_frame")
        if fr is None or not self.__dict__.get("_hip_pending") or torch.is_grad_enabled():
            return None
        vertices, faces, _alpha, _scale = self._hip_inputs()
        return fr[0][k] if fr[1] == _stamp(vertices, faces, _alpha, _scale, self._opacity) else None

    @property
    def get_xyz(self):
        v = self._hip_frame_value(0)
        if v is not None:
            return v
        self._hip_materialize()
        return self._xyz

    @property
    def get_scaling(self):
        v = self._hip_frame_value(1)
        if v is not None:
            return v
        self._hip_materialize()
        act = self.__dict__.get("_hip_activated")
        if act is not None and act[0] is self._scaling:
            return act[2]
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        v = self._hip_frame_value(2)
        if v is not None:
            return v
        self._hip_materialize()
        act = self.__dict__.get("_hip_activated")
        if act is not None and act[1] is self._rotation:
            return act[3]
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_opacity(self):
        v = self._hip_frame_value(3)
        if v is not None:
            return v
        self._hip_materialize()
        cached = self.__dict__.get("_hip_opacity")
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


class HipMeshMixin(_HipGetters):
    """Host class provides: vertices [V,3], faces [F,3], _alpha [F,S,3], _scale [P,1] (attribute name in
    `_hip_scale_attr`); optional `_opacity` [P,1].  `alpha_mode`: "relu" (mesh models) or "softmax" (FLAME)."""

    alpha_mode = "relu"
    _hip_scale_attr = "_scale"

    # ---- deferred K0 (opt-in; games_hip/train.py and bench.py switch it on).  train.py:154-157 calls update_alpha() /
    # prepare_scaling_rot() after every optimizer step and the next thing that happens is render() (train.py:100): with
    # `hip_defer_k0 = True` the two calls only mark the derived attributes stale, and `games_hip.render.render` renders the frame
    # STRAIGHT FROM THE MESH -- the face -> Gaussian arithmetic runs inside the rasterizer's preprocess thread, which also stores
    # xyz / activated scale / unit quaternion / sigmoid opacity for the backward (GmsRasterForwardArgs.mesh_out_*, ABI 6): no K0
    # launch, 84 + 44 bytes per Gaussian less HBM traffic, ONE autograd node from the mesh parameters to the image.  Anything
    # else that asks for the derived values -- the property getters, save_ply, the reference's own render() -- materialises them
    # first with the K0 launch, so every reader sees current values; only code that reads the RAW attributes (`_xyz`,
    # `_scaling`, `_rotation`, `alpha`) directly between an optimizer step and the next render would see the previous step's,
    # which is why the mode is opt-in (the reference's train.py has no such reader: train.py:100-157).
    hip_defer_k0 = False

    def _hip_defer_now(self):
        return (self.hip_defer_k0 and torch.is_grad_enabled() and self.__dict__.get("_hip_tri_external") is None
                and "_xyz" in self.__dict__ and "_scaling" in self.__dict__)          # (the first K0 of a model's life is always eager)

    def _hip_materialize(self):
        if self.__dict__.pop("_hip_pending", None):
            keep, self.hip_defer_k0 = self.hip_defer_k0, False
            try:
                self.update_alpha()
                self.prepare_scaling_rot()
            finally:
                self.hip_defer_k0 = keep

    def _hip_fused_frame(self, xyz, scaling_act, rotation_unit, opacity_act):
        """What a training frame rendered straight from the mesh derived (games_hip.render): served by the getters while the
        inputs are unchanged, so that e.g. an evaluation pass right after a training step launches no K0 either."""
        vertices, faces, _alpha, _scale = self._hip_inputs()
        self.__dict__["_hip_frame"] = ((xyz, scaling_act, rotation_unit, opacity_act), _stamp(vertices, faces, _alpha, _scale, self._opacity))

    # ---- inputs of the op (overridden by the FLAME mixin, whose vertices come out of the FLAME layer)
    def _hip_inputs(self):
        return self.vertices, self.faces, self._alpha, getattr(self, self._hip_scale_attr)

    def _hip_run(self):
        vertices, faces, _alpha, _scale = self._hip_inputs()
        P = int(_alpha.shape[0] * _alpha.shape[1])
        # create_from_pcd calls update_alpha() before `_scale` exists (gaussian_mesh_model.py:78-81,
        # gaussian_flame_model.py:78-82): alpha / xyz do not depend on it, scaling / rotation are not cached then
        have_scale = torch.is_tensor(_scale) and _scale.numel() == P
        scale_in = _scale if have_scale else torch.ones((P, 1), dtype=torch.float32, device=_alpha.device)
        opa = getattr(self, "_opacity", None)
        fuse_opacity = have_scale and torch.is_tensor(opa) and opa.is_cuda and opa.numel() == P
        out = mesh_to_gaussians(vertices, faces, _alpha, scale_in, self.alpha_mode, fused_activations=True,
                                _opacity=opa if fuse_opacity else None)
        alpha, xyz, scaling, rotation, scaling_act, rotation_unit = out[:6]
        self.__dict__["_hip_opacity"] = (opa, opa._version, out[6]) if fuse_opacity else None
        self.__dict__["_hip_cached"] = ((scaling, rotation, scaling_act, rotation_unit),
                                        _stamp(vertices, faces, _alpha, _scale)) if have_scale else None
        return alpha, xyz

    def update_alpha(self):
        if self._hip_defer_now():
            self.__dict__["_hip_pending"] = True          # render() will derive the Gaussians inside the rasterizer (see hip_defer_k0)
            return
        self.__dict__.pop("_hip_pending", None)
        alpha, xyz = self._hip_run()
        self.alpha = alpha
        self._xyz = xyz
        # `triangles` is only read by save_ply and the animated renderers: gathered on first access
        self.__dict__.pop("_hip_tri", None)
        self.__dict__["_hip_tri_external"] = None
        if "triangles" in self.__dict__:           # a save_ply put it there (see save_ply below): keep it fresh
            self.__dict__["triangles"] = self.triangles

    @property
    def triangles(self):
        ext = self.__dict__.get("_hip_tri_external")
        if ext is not None:
            return ext
        tri = self.__dict__.get("_hip_tri")
        vertices, faces = getattr(self, "vertices", None), getattr(self, "faces", None)
        if tri is None and torch.is_tensor(vertices) and torch.is_tensor(faces) and faces.numel():
            with torch.no_grad():
                tri = vertices[faces]
            self.__dict__["_hip_tri"] = tri
        return tri

    @triangles.setter
    def triangles(self, value):
        # a renderer / loader replaced pc.triangles (renderer/gaussian_animated_renderer/__init__.py:72,
        # GaussianMeshModel.load_ply :231)
        self.__dict__["_hip_tri_external"] = value

    def prepare_scaling_rot(self, *unused):
        if self.__dict__.get("_hip_pending") and self._hip_defer_now():
            return
        tri = self.__dict__.get("_hip_tri_external")
        cached = self.__dict__.get("_hip_cached")
        vertices, faces, _alpha, _scale = self._hip_inputs()
        if tri is not None:
            # a renderer replaced pc.triangles (renderer/gaussian_animated_renderer/__init__.py:72-73):
            # derive scale / rotation from those triangles
            _, _, scaling, rotation, scaling_act, rotation_unit = triangles_to_gaussians(
                tri, _alpha, _scale, self.alpha_mode, fused_activations=True)
        elif cached is not None and cached[1] == _stamp(vertices, faces, _alpha, _scale):
            scaling, rotation, scaling_act, rotation_unit = cached[0]
        else:   # no update_alpha() since the inputs changed: recompute from the current tensors, as the reference does
            _, _, scaling, rotation, scaling_act, rotation_unit = mesh_to_gaussians(
                vertices, faces, _alpha, _scale, self.alpha_mode, fused_activations=True)
        self._scaling = scaling
        self._rotation = rotation
        self.__dict__["_hip_activated"] = (scaling, rotation, scaling_act, rotation_unit)

    def save_ply(self, path):
        """The reference's save_ply (gaussian_mesh_model.py:189-207) reads `triangles` out of the instance
        `__dict__`; here it is a lazily gathered property, so materialise it there first (update_alpha keeps it
        fresh from then on)."""
        self._hip_materialize()
        self.__dict__["triangles"] = self.triangles
        return super().save_ply(path)


class HipFlameMixin(HipMeshMixin):
    """GaussianFlameModel: softmax barycentric weights (gaussian_flame_model.py:195), vertices produced by the host
    class's FLAME layer (:196-207, python/torch: linear blend skinning is outside the hot path, SURVEY 8 out-of-scope),
    scale parameter named `_scales` (:42,:82).  One launch derives alpha / xyz / scaling / rotation from them."""

    alpha_mode = "softmax"
    _hip_scale_attr = "_scales"

    def _hip_flame_vertices(self):
        pc = self.point_cloud
        vertices, _ = pc.flame_model(shape_params=self._flame_shape, expression_params=self._flame_exp,
                                     pose_params=self._flame_pose, neck_pose=self._flame_neck_pose,
                                     transl=self._flame_trans)
        return pc.transform_vertices_function(vertices, self._vertices_enlargement)

    def update_alpha(self):
        self.vertices = self._hip_flame_vertices()
        alpha, xyz = self._hip_run()
        self.alpha = alpha
        self._xyz = xyz
        self.__dict__.pop("_hip_tri", None)
        self.__dict__["_hip_tri_external"] = None

    def _hip_defer_now(self):       # (the FLAME layer runs in update_alpha: K0 stays an eager launch)
        return False

    def save_ply(self, path):       # GaussianFlameModel.save_ply does not read `triangles`
        return super(HipMeshMixin, self).save_ply(path)


class HipMultiMeshMixin(_HipGetters):
    """GaussianMultiMeshModel.update_alpha / _calc_xyz / prepare_scaling_rot: the per-mesh python loop and the
    torch.cat of its results become ONE launch over the concatenated meshes (faces re-indexed by the vertex
    offsets, splat ranges as CSR because every mesh may carry a different number of splats per face).
    Host class provides the reference's list attributes: vertices[i], faces[i], _alpha[i] [F_i,S_i,3], _scale[i] [P_i,1]."""

    def _hip_topology(self):
        key = (_stamp(*self.faces), tuple((int(f.shape[0]), int(a.shape[1]), int(v.shape[0]))
                                          for f, a, v in zip(self.faces, self._alpha, self.vertices)))
        cached = self.__dict__.get("_hip_topo")
        if cached is not None and cached[0] == key:
            return cached[1:]
        device = self.vertices[0].device
        faces, counts = [], []
        voff = 0
        for f, a, v in zip(self.faces, self._alpha, self.vertices):
            faces.append(torch.as_tensor(f).to(device).long() + voff)
            counts.append(torch.full((int(f.shape[0]),), int(a.shape[1]), dtype=torch.int64))
            voff += int(v.shape[0])
        counts = torch.cat(counts)
        fso = torch.cat([torch.zeros(1, dtype=torch.int64), counts.cumsum(0)]).to(torch.int32)
        sf = torch.repeat_interleave(torch.arange(counts.numel(), dtype=torch.int32), counts)
        topo = (torch.cat(faces).contiguous(), fso.to(device), sf.to(device))
        self.__dict__["_hip_topo"] = (key,) + topo
        return topo

    def _hip_run(self):
        faces, fso, sf = self._hip_topology()
        V = torch.cat(list(self.vertices))
        A = torch.cat([a.reshape(-1, 3) for a in self._alpha])
        Sc = torch.cat(list(self._scale))
        opa = getattr(self, "_opacity", None)
        fuse_opacity = torch.is_tensor(opa) and opa.is_cuda and opa.numel() == Sc.numel()
        out = mesh_to_gaussians(V, faces, A, Sc, "relu", face_splat_offset=fso, splat_face=sf, fused_activations=True,
                                _opacity=opa if fuse_opacity else None)
        self.__dict__["_hip_opacity"] = (opa, opa._version, out[6]) if fuse_opacity else None
        self.__dict__["_hip_cached"] = (tuple(out[2:6]), _stamp(*self.vertices, *self.faces, *self._alpha, *self._scale))
        return out[0], out[1]

    def update_alpha(self):
        alpha, xyz = self._hip_run()
        sizes = [a.shape[0] * a.shape[1] for a in self._alpha]
        self.alpha = [x.reshape(a.shape) for x, a in zip(torch.split(alpha, sizes), self._alpha)]
        self._xyz = xyz

    def _calc_xyz(self):
        self.update_alpha()

    def prepare_scaling_rot(self, *unused):
        cached = self.__dict__.get("_hip_cached")
        if cached is None or cached[1] != _stamp(*self.vertices, *self.faces, *self._alpha, *self._scale):
            self._hip_run()
            cached = self.__dict__["_hip_cached"]
        scaling, rotation, scaling_act, rotation_unit = cached[0]
        self._scaling, self._rotation = scaling, rotation
        self.__dict__["_hip_activated"] = (scaling, rotation, scaling_act, rotation_unit)


# ---------------------------------------------------------------------------------------------------------------
# Stand-alone hosts (no dependency on the reference tree) used by bench.py / the GPU tests
class _StandaloneBase:
    def __init__(self, sh_degree: int = 3):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        self.optimizer = None

    @property
    def get_xyz(self):
        return self._xyz

    def oneupSHdegree(self):        # scene/gaussian_model.py:117-119
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    # ---- point_cloud.ply in the reference's layout (scene/gaussian_model.py:177-217 construct_list_of_attributes + save_ply,
    # :229-268 load_ply), through the plyfile stand-in
    def construct_list_of_attributes(self):
        names = ["x", "y", "z", "nx", "ny", "nz"]
        names += [f"f_dc_{i}" for i in range(self._features_dc.shape[1] * self._features_dc.shape[2])]
        names += [f"f_rest_{i}" for i in range(self._features_rest.shape[1] * self._features_rest.shape[2])]
        names += ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]
        return names

    def _save_point_cloud(self, path):
        import os
        import numpy as np
        from ._plyfile_compat import PlyData, PlyElement
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        xyz = self._xyz.detach().cpu().numpy()
        cols = [xyz, np.zeros_like(xyz),
                self._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy(),
                self._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy(),
                self._opacity.detach().cpu().numpy(), self._scaling.detach().cpu().numpy(), self._rotation.detach().cpu().numpy()]
        elements = np.empty(xyz.shape[0], dtype=[(a, "f4") for a in self.construct_list_of_attributes()])
        elements[:] = list(map(tuple, np.concatenate(cols, axis=1)))
        PlyData([PlyElement.describe(elements, "vertex")]).write(path)

    def _load_point_cloud(self, path, device):
        """-> dict of float32 device tensors: xyz [P,3], features_dc [P,1,3], features_rest [P,15,3], opacity [P,1],
        scaling [P,3], rotation [P,4] (the columns of scene/gaussian_model.py:229-268)."""
        import numpy as np
        from ._plyfile_compat import PlyData
        el = PlyData.read(path).elements[0]
        col = lambda n: np.asarray(el[n], dtype=np.float32)
        P = el.count
        f_dc = np.stack([col(f"f_dc_{i}") for i in range(3)], axis=1).reshape(P, 3, 1)
        rest_names = sorted([p.name for p in el.properties if p.name.startswith("f_rest_")], key=lambda x: int(x.split("_")[-1]))
        assert len(rest_names) == 3 * (self.max_sh_degree + 1) ** 2 - 3
        f_rest = np.stack([col(n) for n in rest_names], axis=1).reshape(P, 3, (self.max_sh_degree + 1) ** 2 - 1)
        scale_names = sorted([p.name for p in el.properties if p.name.startswith("scale_")], key=lambda x: int(x.split("_")[-1]))
        rot_names = sorted([p.name for p in el.properties if p.name.startswith("rot_")], key=lambda x: int(x.split("_")[-1]))
        t = lambda a: torch.tensor(a, dtype=torch.float, device=device).contiguous()
        return dict(xyz=t(np.stack([col("x"), col("y"), col("z")], axis=1)), features_dc=t(np.transpose(f_dc, (0, 2, 1))),
                    features_rest=t(np.transpose(f_rest, (0, 2, 1))), opacity=t(col("opacity")[:, None]),
                    scaling=t(np.stack([col(n) for n in scale_names], axis=1)), rotation=t(np.stack([col(n) for n in rot_names], axis=1)))

    def _make_optimizer(self, groups, fused):
        if fused:       # one HIP launch per step (csrc/adam.hip); same state layout as torch.optim.Adam
            from .optim import FusedAdam
            self.optimizer = FusedAdam(groups, lr=0.0, eps=1e-15)
        else:
            self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)


class HipGaussianMeshModel(HipMeshMixin, _StandaloneBase):
    """gs_mesh (games/mesh_splatting/scene/gaussian_mesh_model.py) on the fused op."""

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
        """Parameter groups of gaussian_mesh_model.py:171-183."""
        self._make_optimizer([
            {"params": [self.vertices], "lr": vertices_lr, "name": "vertices"},
            {"params": [self._alpha], "lr": alpha_lr, "name": "alpha"},
            {"params": [self._features_dc], "lr": feature_lr, "name": "f_dc"},
            {"params": [self._features_rest], "lr": feature_lr / 20.0, "name": "f_rest"},
            {"params": [self._opacity], "lr": opacity_lr, "name": "opacity"},
            {"params": [self._scale], "lr": scaling_lr, "name": "scaling"},
        ], fused)

    # ---- checkpoints: the reference's on-disk format (scene/gaussian_model.py:177-268 point_cloud.ply +
    # games/mesh_splatting/scene/gaussian_mesh_model.py:189-222 model_params.pt), through the plyfile stand-in
    def save_ply(self, path):
        self.update_alpha()
        self.prepare_scaling_rot()
        self._save_point_cloud(path)
        torch.save({"_alpha": self._alpha, "_scale": self._scale, "point_cloud": None, "triangles": self.triangles,
                    "vertices": self.vertices, "faces": self.faces}, path.replace("point_cloud.ply", "model_params.pt"))

    def load_ply(self, path, device="cuda"):
        pc = self._load_point_cloud(path, device)
        par = lambda a: nn.Parameter(a.requires_grad_(True))
        self._features_dc = par(pc["features_dc"])
        self._features_rest = par(pc["features_rest"])
        self._opacity = par(pc["opacity"])
        params = torch.load(path.replace("point_cloud.ply", "model_params.pt"), map_location=device, weights_only=False)
        self.vertices = nn.Parameter(params["vertices"].detach().to(device))
        self.faces = params["faces"].to(device)
        self._alpha = nn.Parameter(params["_alpha"].detach().to(device))
        self._scale = nn.Parameter(params["_scale"].detach().to(device))
        self.active_sh_degree = self.max_sh_degree
        self.update_alpha()
        self.prepare_scaling_rot()


class HipGaussianMultiMeshModel(HipMultiMeshMixin, _StandaloneBase):
    """gs_multi_mesh (BASELINE config 4): several meshes, each with its own splats-per-face, one set of SH /
    opacity tensors over the concatenation (gaussian_multi_mesh_model.py:48-97)."""

    @classmethod
    def from_scenes(cls, scenes, device="cuda"):
        m = cls(3)
        m.active_sh_degree = scenes[0].active_sh_degree
        par = lambda t: nn.Parameter(t.to(device).float().contiguous())
        m.vertices = [par(s.vertices) for s in scenes]
        m.faces = [s.faces.to(device) for s in scenes]
        m._alpha = [par(s._alpha) for s in scenes]
        m._scale = [par(s._scale) for s in scenes]
        m._opacity = par(torch.cat([s._opacity for s in scenes]))
        m._features_dc = par(torch.cat([s._features_dc for s in scenes]))
        m._features_rest = par(torch.cat([s._features_rest for s in scenes]))
        m.update_alpha()
        m.prepare_scaling_rot()
        return m

    def parameters(self):
        return [*self.vertices, *self._alpha, self._features_dc, self._features_rest, self._opacity, *self._scale]

    def training_setup(self, vertices_lr=0.0, alpha_lr=0.001, feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005,
                       fused=True):
        """Parameter groups of gaussian_multi_mesh_model.py:221-236."""
        self._make_optimizer([
            {"params": list(self._alpha), "lr": alpha_lr, "name": "alpha"},
            {"params": list(self.vertices), "lr": vertices_lr, "name": "vertices"},
            {"params": [self._features_dc], "lr": feature_lr, "name": "f_dc"},
            {"params": [self._features_rest], "lr": feature_lr / 20.0, "name": "f_rest"},
            {"params": [self._opacity], "lr": opacity_lr, "name": "opacity"},
            {"params": list(self._scale), "lr": scaling_lr, "name": "scaling"},
        ], fused)


class _SyntheticFlameLayer:
    """Stand-in for the licensed FLAME layer (games/flame_splatting/FLAME, absent: needs smplx + the model file):
    a differentiable vertex generator with the same call signature and return shape ([1,V,3], landmarks) --
    template + expression-weighted blend shapes + a global rotation about z by pose[0,0] + translation."""

    def __init__(self, template, n_exp=4, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.template = template
        self.blend = 0.02 * torch.randn(n_exp, *template.shape, generator=g).to(template.device)

    def __call__(self, shape_params=None, expression_params=None, pose_params=None, neck_pose=None, transl=None):
        v = self.template + torch.einsum("e,evk->vk", expression_params.reshape(-1)[: self.blend.shape[0]], self.blend)
        a = pose_params.reshape(-1)[0]
        c, s = torch.cos(a), torch.sin(a)
        R = torch.stack([torch.stack([c, -s, torch.zeros_like(c)]), torch.stack([s, c, torch.zeros_like(c)]),
                         torch.stack([torch.zeros_like(c), torch.zeros_like(c), torch.ones_like(c)])])
        v = v @ R.T + transl.reshape(1, 3)
        return v[None], None


def _squeeze_and_enlarge(vertices, enlargement):
    """transform_vertices_function of games/flame_splatting/scene/dataset_readers.py:41-46 without the axis swap (a module-level
    function, not a lambda: `point_cloud` is pickled into flame_params.pt as the reference does with its FLAMEPointCloud)."""
    return torch.squeeze(vertices, 0) * enlargement


class _FlameCloud:
    """The attributes of FLAMEPointCloud (games/flame_splatting/utils/graphics_utils.py) the model reads."""

    def __init__(self, flame_model, transform_vertices_function):
        self.flame_model, self.transform_vertices_function = flame_model, transform_vertices_function


class HipGaussianFlameModel(HipFlameMixin, _StandaloneBase):
    """gs_flame (BASELINE config 5) with a synthetic vertex generator in place of the FLAME layer: per-frame
    vertex animation + on-device re-derivation of the face-local rotation / scale."""

    @classmethod
    def from_scene(cls, scene, device="cuda", enlargement=1.0):
        m = cls(3)
        m.active_sh_degree = scene.active_sh_degree
        par = lambda t: nn.Parameter(t.to(device).float().contiguous())
        template = scene.vertices.to(device).float()
        m.point_cloud = _FlameCloud(_SyntheticFlameLayer(template), _squeeze_and_enlarge)
        m.faces = scene.faces.to(device)
        m._flame_shape = par(torch.zeros(1, 4))
        m._flame_exp = par(torch.zeros(1, 4))
        m._flame_pose = par(torch.zeros(1, 6))
        m._flame_neck_pose = par(torch.zeros(1, 3))
        m._flame_trans = par(torch.zeros(1, 3))
        m._vertices_enlargement = par(torch.full_like(template, float(enlargement)))
        m._alpha = par(scene._alpha)
        m._scales = par(scene._scale)
        m._opacity = par(scene._opacity)
        m._features_dc = par(scene._features_dc)
        m._features_rest = par(scene._features_rest)
        m.update_alpha()
        m.prepare_scaling_rot()
        return m

    def parameters(self):
        return [self._flame_exp, self._flame_pose, self._flame_trans, self._vertices_enlargement, self._alpha,
                self._features_dc, self._features_rest, self._opacity, self._scales]

    def training_setup(self, flame_shape_lr=0.01, flame_exp_lr=0.001, flame_pose_lr=0.001, flame_neck_pose_lr=0.001,
                       flame_trans_lr=0.001, vertices_enlargement_lr=0.0002, alpha_lr=0.001, feature_lr=0.0025, opacity_lr=0.05,
                       scaling_lr=0.005, fused=True):
        """Parameter groups of gaussian_flame_model.py:209-228, in its order, with the defaults of `OptimizationParamsFlame`
        (arguments_games/__init__.py:30-47)."""
        self._make_optimizer([
            {"params": [self._flame_shape], "lr": flame_shape_lr, "name": "shape"},
            {"params": [self._flame_exp], "lr": flame_exp_lr, "name": "expression"},
            {"params": [self._flame_pose], "lr": flame_pose_lr, "name": "pose"},
            {"params": [self._flame_neck_pose], "lr": flame_neck_pose_lr, "name": "neck_pose"},
            {"params": [self._flame_trans], "lr": flame_trans_lr, "name": "transl"},
            {"params": [self._vertices_enlargement], "lr": vertices_enlargement_lr, "name": "vertices_enlargement"},
            {"params": [self._alpha], "lr": alpha_lr, "name": "alpha"},
            {"params": [self._features_dc], "lr": feature_lr, "name": "f_dc"},
            {"params": [self._features_rest], "lr": feature_lr / 20.0, "name": "f_rest"},
            {"params": [self._opacity], "lr": opacity_lr, "name": "opacity"},
            {"params": [self._scales], "lr": scaling_lr, "name": "scaling"},
        ], fused)

    # ---- checkpoints: point_cloud.ply + flame_params.pt (games/flame_splatting/scene/gaussian_flame_model.py:232-265)
    FLAME_ATTRS = ("_flame_shape", "_flame_exp", "_flame_pose", "_flame_neck_pose", "_flame_trans", "_vertices_enlargement",
                   "faces", "alpha", "point_cloud")                 # the reference's `flame_additional_attrs` (:238-244), same order
    FLAME_EXTRA_ATTRS = ("_alpha", "_scales")                       # not in the reference's file: see load_ply

    def save_ply(self, path):
        """`save_ply` of the reference (:232-251): refresh alpha / scaling / rotation, write point_cloud.ply, then a dict of
        the FLAME attributes next to it.  Two extra keys (`_alpha`, `_scales`: the RAW barycentric logits and the per-splat
        scale multipliers) ride along -- the reference's own `load_ply` ignores unknown keys; without them a loaded model can
        only replay the PLY's scaling / rotation, not re-derive them per animated frame (BASELINE
        config 5)."""
        self.update_alpha()
        self.prepare_scaling_rot()
        self._save_point_cloud(path)
        save_dict = {k: getattr(self, k) for k in self.FLAME_ATTRS + self.FLAME_EXTRA_ATTRS}
        torch.save(save_dict, path.replace("point_cloud.ply", "flame_params.pt"))

    def load_ply(self, path, device="cuda"):
        """`load_ply` of the reference (:253-265): the base model's PLY columns (`_xyz`, features, `_opacity`, `_scaling`,
        `_rotation` as Parameters, scene/gaussian_model.py:229-268) and the nine FLAME attributes.  A file written by the
        reference has no `_alpha` / `_scales`: the model then holds exactly what the reference's holds after loading --
        `alpha` (activated) for `flame_render`'s xyz, scaling / rotation from the PLY, `vertices = None`
        (renderer/flame_gaussian_renderer/__init__.py:59-80).  A file written by `save_ply` above restores the raw
        parameters as well and re-derives everything on the device."""
        pc = self._load_point_cloud(path, device)
        par = lambda a: nn.Parameter(a.requires_grad_(True))
        self._xyz = par(pc["xyz"])
        self._features_dc = par(pc["features_dc"])
        self._features_rest = par(pc["features_rest"])
        self._opacity = par(pc["opacity"])
        self._scaling = par(pc["scaling"])
        self._rotation = par(pc["rotation"])
        self.active_sh_degree = self.max_sh_degree
        params = torch.load(path.replace("point_cloud.ply", "flame_params.pt"), map_location=device, weights_only=False)
        for k in self.FLAME_ATTRS:
            setattr(self, k, params[k])
        self.vertices = None
        self.__dict__["_hip_cached"] = None
        self.__dict__["_hip_opacity"] = None
        # the getters serve exp(_scaling) / normalize(_rotation) of the PLY columns until something re-derives them
        self.__dict__["_hip_activated"] = None
        if all(k in params for k in self.FLAME_EXTRA_ATTRS):
            self._alpha = nn.Parameter(params["_alpha"].detach().to(device))
            self._scales = nn.Parameter(params["_scales"].detach().to(device))
            self.update_alpha()
            self.prepare_scaling_rot()


_MIXINS = {"gs_mesh": HipMeshMixin, "gs_multi_mesh": HipMultiMeshMixin, "gs_flame": HipFlameMixin}


def install(games_module=None):
    """Swap the fused op into the reference's registries (games/__init__.py:35-51): `gaussianModel` (train.py) and
    `gaussianModelRender` (scripts/render.py:22,41) for gs_mesh, gs_multi_mesh and gs_flame.  Every model keeps its
    class, dataset reader, optimizer groups and PLY I/O; only update_alpha / prepare_scaling_rot (+ the fused property
    getters) are overridden.  Call after `import games` in an environment that has the reference on sys.path.
    Returns {name: patched class}; `uninstall(games_module, returned)` restores the originals."""
    if games_module is None:
        import games as games_module  # type: ignore
    out = {}
    for name, mixin in _MIXINS.items():
        base = games_module.gaussianModel[name]
        if issubclass(base, mixin):            # already installed
            out[name] = base
            continue
        cls = type("Hip" + base.__name__, (mixin, base), {"_hip_base": base})
        for registry in (games_module.gaussianModel, getattr(games_module, "gaussianModelRender", {})):
            if registry.get(name) is base:
                registry[name] = cls
        out[name] = cls
    return out


def uninstall(games_module, installed):
    for name, cls in installed.items():
        base = getattr(cls, "_hip_base", None)
        if base is None:
            continue
        for registry in (games_module.gaussianModel, getattr(games_module, "gaussianModelRender", {})):
            if registry.get(name) is cls:
                registry[name] = base
