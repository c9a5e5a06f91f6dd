"""Animated render drivers: the loops of scripts/render_time_animated.py:68-87 and
scripts/render_flame.py:29-60 on the fused op -- per frame the mesh is deformed, the face-local rotation / scale are
re-derived on the device (one K0 launch on the explicit triangles) and the frame is rasterized; nothing leaves the GPU
unless `out_dir` is given.

    frames = render_time_animated(model, views, pipe, bg, transform=transform_hotdog_fly)

The vertex transforms are the reference's (scripts/render_time_animated.py:28-66) with the reference's semantics:
`transform_hotdog_fly` and `make_smaller` return a deformed COPY of the rest pose; `transform_ficus_sinus`,
`transform_ficus_pot` and `transform_ship_sinus` write IN PLACE into the tensor they are given, so -- as in the reference,
which hands them `gaussians.vertices` itself -- their deformation ACCUMULATES from frame to frame.  The one deviation: the
driver below hands them a working copy made once per call, so the model's own `vertices` parameter is left untouched
(the reference's loop leaves the model deformed after rendering).
"""
from __future__ import annotations

import math
import os
from typing import Callable, Iterable, List, Optional

import torch

from .render import render, render_animated


def _sel(idxs):
    return slice(None) if idxs is None else idxs


def transform_hotdog_fly(vertices, t, idxs=None):          # scripts/render_time_animated.py:35-41 (copy of the rest pose)
    v = vertices.clone()
    v[:, 2] += float(t) * (vertices[:, 1] ** 2 + vertices[:, 1] ** 2) ** 0.5 * 0.01
    return v


def transform_ship_sinus(vertices, t, idxs=None):          # :52-55 (in place: accumulates over the frames)
    f = math.sin(float(t)) * 0.5
    vertices[:, 2] += 0.05 * torch.sin(vertices[:, 0] * math.pi + f)
    return vertices


def transform_ficus_sinus(vertices, t, idxs=None):         # :28-32 (two sinus terms, in place: accumulates)
    i = _sel(idxs)
    vertices[i, 2] += 0.005 * torch.sin(vertices[i, 0] * 2 * math.pi + float(t))
    vertices[i, 2] += 0.005 * torch.sin(vertices[i, 1] * 5 * math.pi + float(t))
    return vertices


def transform_ficus_pot(vertices, t, idxs=None):           # :44-49 (in place: accumulates)
    i = _sel(idxs)
    if float(t) > 8 * math.pi:
        vertices[i, 2] += 0.005 * torch.sin(vertices[i, 1] * 5 * math.pi + float(t))
    else:
        vertices[i, 2] -= (0.005 + float(t)) * (vertices[i, 0] / 10) ** 2
    return vertices


def make_smaller(vertices, t, idxs=None):                  # :58-62
    return (math.sin(float(t)) + 1.0) * vertices


def _save(image: torch.Tensor, path: str) -> None:
    """torchvision.utils.save_image(rendering, path) of the reference (PNG, [0,1] clamp, 8 bit)."""
    from PIL import Image
    a = (image.detach().clamp(0, 1) * 255.0 + 0.5).to(torch.uint8).permute(1, 2, 0).cpu().numpy()
    Image.fromarray(a).save(path)


@torch.no_grad()
def render_time_animated(gaussians, views: Iterable, pipeline, background: torch.Tensor,
                         transform: Callable = transform_hotdog_fly, idxs=None, out_dir: Optional[str] = None,
                         t_max: float = 10 * math.pi) -> List[torch.Tensor]:
    """scripts/render_time_animated.py:68-87.  Returns the rendered frames (device tensors [3,H,W])."""
    views = list(views)
    ts = torch.linspace(0, t_max, max(len(views), 1))
    vertices = gaussians.vertices.detach().clone()      # working copy: the in-place transforms accumulate in it, as in the reference
    faces = gaussians.faces.long()
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
    frames = []
    for k, view in enumerate(views):
        new_vertices = transform(vertices, ts[k], idxs)
        triangles = new_vertices[faces].float()
        img = render_animated(idxs, triangles, view, gaussians, pipeline, background)["render"]
        frames.append(img)
        if out_dir:
            _save(img, os.path.join(out_dir, f"{k:05d}.png"))
    return frames


@torch.no_grad()
def render_flame_animated(gaussians, views: Iterable, pipeline, background: torch.Tensor, drive: Callable,
                          out_dir: Optional[str] = None) -> List[torch.Tensor]:
    """scripts/render_flame.py:29-60 shape: per frame `drive(gaussians, k)` edits the FLAME parameters in place
    (expression / pose / neck / translation), update_alpha() runs the FLAME layer + ONE K0 launch, then the plain render."""
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
    frames = []
    for k, view in enumerate(views):
        drive(gaussians, k)
        gaussians.update_alpha()
        gaussians.prepare_scaling_rot()
        img = render(view, gaussians, pipeline, background)["render"]
        frames.append(img)
        if out_dir:
            _save(img, os.path.join(out_dir, f"{k:05d}.png"))
    return frames
