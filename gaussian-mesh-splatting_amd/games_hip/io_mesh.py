"""Mesh input of the mesh-bound models without the absent third-party readers.

  load_obj()                 what `trimesh.load(f'{path}/mesh.obj', force='mesh')` gives the reference at
                             games/mesh_splatting/scene/dataset_readers.py:49 (and scripts/render_time_animated.py:100):
                             `.vertices` [V,3] float64, `.faces` [F,3] int64 -- Wavefront OBJ `v` / `f` records, polygons
                             fan-triangulated, negative (relative) indices, `v/vt/vn` index triplets.
  save_obj()                 the inverse (scripts/save_pseudomesh.py writes pseudo-meshes this way through trimesh).
  transform_vertices_function  dataset_readers.py:31-37: (x, y, z) -> (x, -z, y) * c, the Blender -> world axis swap.
  mesh_point_cloud()         dataset_readers.py:62-92: `num_splats` random barycentric points per face ->
                             the fields of `MeshPointCloud` (games/mesh_splatting/utils/graphics_utils.py:19-27).
Host-side parsing only (numpy); tensors move to the GPU in the model's create_from_pcd.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np
import torch

C0 = 0.28209479177387814


@dataclass
class TriMesh:
    """The two attributes of trimesh.Trimesh the reference reads."""
    vertices: np.ndarray      # [V,3] float64
    faces: np.ndarray         # [F,3] int64

    @property
    def triangles(self):      # scripts/render_from_object.py:34 reads mesh_scene.triangles
        return self.vertices[self.faces]


def load_obj(path: str) -> TriMesh:
    verts, faces = [], []
    with open(path, "r", errors="replace") as f:
        for line in f:
            if line.startswith("v "):
                p = line.split()
                verts.append((float(p[1]), float(p[2]), float(p[3])))
            elif line.startswith("f "):
                idx = []
                for tok in line.split()[1:]:
                    i = int(tok.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)          # 1-based, or relative to the vertices read so far
                for k in range(1, len(idx) - 1):                            # fan triangulation of polygons
                    faces.append((idx[0], idx[k], idx[k + 1]))
    v = np.asarray(verts, dtype=np.float64).reshape(-1, 3)
    fa = np.asarray(faces, dtype=np.int64).reshape(-1, 3)
    if fa.size and (fa.min() < 0 or fa.max() >= len(v)):
        raise ValueError(f"{path}: face index out of range")
    return TriMesh(v, fa)


def save_obj(path: str, vertices, faces) -> None:
    v = np.asarray(vertices.detach().cpu() if torch.is_tensor(vertices) else vertices, dtype=np.float64)
    f = np.asarray(faces.detach().cpu() if torch.is_tensor(faces) else faces, dtype=np.int64)
    with open(path, "w") as out:
        for x, y, z in v:
            out.write(f"v {x:.9g} {y:.9g} {z:.9g}\n")
        for a, b, c in f + 1:
            out.write(f"f {a} {b} {c}\n")


def transform_vertices_function(vertices: torch.Tensor, c: float = 1) -> torch.Tensor:
    vertices = vertices[:, [0, 2, 1]]
    vertices[:, 1] = -vertices[:, 1]
    vertices = vertices * c
    return vertices


@dataclass
class MeshPointCloud:
    """Same fields as games/mesh_splatting/utils/graphics_utils.py:19-27 (a NamedTuple there)."""
    alpha: torch.Tensor
    points: torch.Tensor
    colors: np.ndarray
    normals: np.ndarray
    vertices: torch.Tensor
    faces: np.ndarray
    transform_vertices_function: Optional[Callable]
    triangles: torch.Tensor


def mesh_point_cloud(mesh: TriMesh, num_splats: int, seed: Optional[int] = None, transform=transform_vertices_function) -> MeshPointCloud:
    """dataset_readers.py:49-92: transformed vertices, `num_splats` uniform-random alpha rows per face (NOT normalised:
    update_alpha does relu + L1), points = alpha @ triangles, near-black random colours."""
    g = torch.Generator().manual_seed(seed) if seed is not None else None
    vertices = transform(torch.tensor(mesh.vertices)) if transform is not None else torch.tensor(mesh.vertices)
    faces = mesh.faces
    triangles = vertices[torch.tensor(faces).long()].float()
    F = triangles.shape[0]
    alpha = torch.rand(F, num_splats, 3, generator=g)
    xyz = torch.matmul(alpha, triangles).reshape(F * num_splats, 3)
    rng = np.random.default_rng(seed)
    shs = rng.random((F * num_splats, 3)) / 255.0
    return MeshPointCloud(alpha=alpha, points=xyz, colors=shs * C0 + 0.5, normals=np.zeros((F * num_splats, 3)),
                          vertices=vertices, faces=faces, transform_vertices_function=transform, triangles=triangles)
