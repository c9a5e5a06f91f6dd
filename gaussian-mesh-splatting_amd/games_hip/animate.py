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

from .render import _fused_frame_ok, render, render_animated, render_mesh_frame


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


def render_frame(vertices: torch.Tensor, faces: torch.Tensor, view, gaussians, pipeline, background: torch.Tensor):
    """One animated frame from deformed vertices.  Forward-only frames of a model the fused path supports go mesh -> image in the
    rasterizer's own launches (`render_mesh_frame`: no `vertices[faces]` gather, no K0 launch, no xyz / scale / rotation tensors);
    anything else takes the reference's route, `render_animated(None, vertices[faces], ...)`.  Same image bit for bit."""
    if _fused_frame_ok(gaussians, pipeline, None):
        return render_mesh_frame(vertices, faces, view, gaussians, pipeline, background)
    return render_animated(None, vertices[faces].float(), view, gaussians, pipeline, background)


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
        img = render_frame(new_vertices, faces, view, gaussians, pipeline, background)["render"]
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


class GraphedAnimation:
    """The animated render loop of scripts/render_time_animated.py:68-87 as a replayed hipGraph.

    One frame = deformed triangles -> fused face->Gaussian op -> rasterizer forward.  After `warmup` ordinary frames on a private
    stream (every allocation, capacity hint and library-owned buffer of the shape in place) one frame is CAPTURED with
    `torch.cuda.graph`; inside the capture the rasterizer runs in its launches-only form (`GmsRasterForwardArgs.no_host_wait`,
    include/gmsplat.h: no host wait for the instance count).  `render(triangles)` then copies the triangles into the graph's static
    input, replays the graph and returns the static output image: no Python between the kernels, no host synchronisation.

    A replayed frame keeps the CAPTURED capacity and launch sizes.  `status()` reads the frame's own counts back from the device
    (one small copy: call it every frame or every few, as the application likes): if a frame had more (Gaussian, tile) instances
    than the captured binning capacity or more work units than the captured launches it is incomplete -- `render(...,
    check=True)` re-captures with the frame's counts and renders it again.

        anim = GraphedAnimation(gaussians, view, pipeline, background)
        for k in range(n_frames):
            img = anim.render(transform(vertices, t[k])[faces].float(), check=True)
    """

    def __init__(self, gaussians, view, pipeline, background: torch.Tensor, warmup: int = 3):
        import diff_gaussian_rasterization as dgr
        if dgr._C is None:
            raise NotImplementedError("GraphedAnimation needs the _C extension module")
        self._dgr = dgr
        self.pc, self.view, self.pipe, self.bg, self.warmup = gaussians, view, pipeline, background, int(warmup)
        self.stream = torch.cuda.Stream(device=background.device)
        self.graph = None
        self.static_tri = None
        self.out = None
        self.captures = 0
        self._image_scratch, self._counts_off, self.capacity, self.launched_units = None, 0, 0, 0

    @torch.no_grad()
    def _capture(self, triangles: torch.Tensor, slack: float = 1.0) -> None:
        dgr = self._dgr
        dev = triangles.device
        self.static_tri = triangles.detach().clone()
        W, H = int(self.view.image_width), int(self.view.image_height)
        P = int(self.pc._alpha.shape[0] * self.pc._alpha.shape[1])
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream):
            for _ in range(max(self.warmup, 1)):          # ordinary frames: hints, pools, the library's per-stream buffers
                render_animated(None, self.static_tri, self.view, self.pc, self.pipe, self.bg)
            if slack > 1.0:                                # a re-capture after an overflow: leave room above this frame's count
                n = int(dgr.last_stats()["num_rendered"])
                dgr.set_capacity_hint(dev.index if dev.index is not None else torch.cuda.current_device(), W, H, P, int(n * slack))
                render_animated(None, self.static_tri, self.view, self.pc, self.pipe, self.bg)
            self.stream.synchronize()
            kept = bool(getattr(dgr, "_keep_buffers", False))      # (the caller's own setting is restored below)
            dgr.keep_buffers(True)                         # the captured frame's image scratch holds its counts: keep the handle
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.out = render_animated(None, self.static_tri, self.view, self.pc, self.pipe, self.bg)["render"]
            st = dict(dgr._C.last_stats())
            if not kept:
                dgr.keep_buffers(False)
        self._image_scratch = st["image"]
        self.capacity = int(st["capacity_hint"])
        self.launched_units = int(dgr._C.last_launched_units())
        self._counts_off = int(dgr._C.image_counts_offset(W, H))
        self.captures += 1
        torch.cuda.current_stream(dev).wait_stream(self.stream)

    def status(self) -> dict:
        """Counts of the most recent replayed frame (device -> host copy of 16 bytes) and whether it fitted the capture."""
        if self._image_scratch is None:          # nothing captured yet
            return {"num_rendered": 0, "deepest_tile": 0, "num_units": 0, "segment_length": 0, "capacity": 0, "launched_units": 0,
                    "complete": True}
        c = self._image_scratch[self._counts_off:self._counts_off + 16].view(torch.int32).cpu()
        n, deepest, units, L = (int(x) & 0xffffffff for x in c)
        return {"num_rendered": n, "deepest_tile": deepest, "num_units": units, "segment_length": L, "capacity": self.capacity,
                "launched_units": self.launched_units, "complete": n <= self.capacity and units <= self.launched_units}

    @torch.no_grad()
    def render(self, triangles: torch.Tensor, check: bool = True) -> torch.Tensor:
        """The frame for `triangles` [F,3,3].  Returns the graph's STATIC output tensor (overwritten by the next call: clone it to
        keep it).  `check` (default True: a frame that outgrew the capture would otherwise come back silently incomplete): verify the
        frame fitted the captured sizes -- a 16-byte read-back, i.e. one stream synchronisation -- and, if it did not, re-capture at
        1.5x its count and redo it.  Pass False and poll `status()` every few frames where the synchronisation matters."""
        dev = triangles.device
        if self.graph is None or triangles.shape != self.static_tri.shape:
            self._capture(triangles)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream):
            self.static_tri.copy_(triangles)
            self.graph.replay()
            if check and not self.status()["complete"]:
                self._capture(triangles, slack=1.5)
                self.static_tri.copy_(triangles)
                self.graph.replay()
        torch.cuda.current_stream(dev).wait_stream(self.stream)
        return self.out
