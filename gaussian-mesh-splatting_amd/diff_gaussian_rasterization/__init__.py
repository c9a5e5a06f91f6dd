"""Drop-in `diff_gaussian_rasterization` for GaMeS on MI355X (gfx950).

Same import surface as the package the reference imports at
renderer/gaussian_renderer/__init__.py:14 (and the three sibling renderers):

    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

  * `GaussianRasterizationSettings`: the 13-field NamedTuple built by keyword at
    renderer/gaussian_renderer/__init__.py:43-57.
  * `GaussianRasterizer(raster_settings)`: nn.Module whose forward takes
    (means3D, means2D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp) and returns
    the 3-tuple (color[3,H,W], radii[P] int32, invdepth[1,H,W]) unpacked at
    renderer/gaussian_renderer/__init__.py:94-102.
  * autograd contract: forward inputs (means3D, means2D, sh, colors_precomp, opacities, scales,
    rotations, cov3Ds_precomp, raster_settings); backward returns nine gradients in that order
    (train.py:108 drives it through loss.backward()).  `means2D.grad` receives the NDC-scaled
    screen-space gradient that densification reads (scene/gaussian_model.py:416-418).

All arithmetic runs in hand-written HIP kernels (libgmsplat.so, C ABI in include/gmsplat.h);
torch only provides device memory, the current stream and autograd plumbing.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import NamedTuple, Optional

import torch
import torch.nn as nn

from . import _lib

# Two bindings of the same C ABI (include/gmsplat.h, libgmsplat.so):
#   `_C`     the PyTorch-ROCm extension module (csrc/torch_binding.cpp): upstream's three entry points plus the autograd
#            node in C++ -- the default, one Python call per render;
#   ctypes   `_lib.py`: the raw C ABI driven from Python (what INTEGRATION.md shows a maintainer), GMS_BINDING=ctypes.
# Both reach the same kernels; neither has a CPU path.
_BINDING = os.environ.get("GMS_BINDING", "auto")
try:
    if _BINDING == "ctypes":
        raise ImportError("GMS_BINDING=ctypes")
    from . import _C  # noqa: F401
except ImportError as _e:          # not built (make -C csrc) or disabled: the ctypes binding serves every call
    if _BINDING == "torch":
        raise
    _C = None

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "last_stats", "SplitSH", "keep_buffers",
           "set_capacity_hint", "clear_capacity_hints"]


class SplitSH:
    """SH coefficients kept as the reference stores them: `_features_dc` [P,1,3] and `_features_rest` [P,M-1,3]
    (scene/gaussian_model.py:107-111 concatenates them into a fresh 57.6 MB tensor every iteration, and autograd
    splits the gradient back with two more copies).  A model's `get_features` may return this object instead
    of the concatenation: `GaussianRasterizer` reads both blocks in place and writes both gradients in place.
    Any other consumer (e.g. the reference's `convert_SHs_python` branch, renderer/gaussian_renderer/__init__.py:83)
    can treat it as the concatenated tensor: attribute access falls through to a lazily built `torch.cat`."""

    def __init__(self, dc: torch.Tensor, rest: torch.Tensor):
        self.dc, self.rest = dc, rest
        self._full = None

    def full(self) -> torch.Tensor:
        if self._full is None:
            self._full = torch.cat((self.dc, self.rest), dim=1)
        return self._full

    @property
    def shape(self):
        return torch.Size((self.dc.shape[0], self.dc.shape[1] + self.rest.shape[1], self.dc.shape[2]))

    def __getattr__(self, name):          # only reached for attributes SplitSH itself does not define
        return getattr(self.full(), name)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    antialiasing: bool


# (device index, W, H, P) -> slowly decaying maximum of the instances rendered by recent calls: lets the next call
# size the binning buffer up-front and enqueue the whole forward without a pipeline bubble.  Training visits cameras
# in random order, so the estimate follows the heaviest recent view, not just the last one (an overflow re-runs the
# tail of the pipeline; memory is 64 bytes per instance).
_capacity_cache = {}
_last_stats = {}


def _quantize_capacity(x: int) -> int:
    """Round a capacity hint UP to four significant bits (steps of 6-12 %): the binning buffer is sized by the hint, and a size
    that creeps up frame by frame (an animated mesh growing 0.5 % per frame) would be a fresh hipMalloc in the caching
    allocator on every frame.  Same rule as torch_binding.cpp::quantize_capacity."""
    if x < 16:
        return x
    s = x.bit_length() - 1 - 3
    return ((x + (1 << s) - 1) >> s) << s


def set_capacity_hint(device_index: int, width: int, height: int, P: int, value: int) -> None:
    """Overwrite the learnt instance count the next forward of this (device, W, H, P) sizes its binning buffer from
    (x1.25 + 4096); `value` < 0 forgets it.  Tests use it to force the overflow re-run."""
    if _C is not None:
        _C.set_capacity(int(device_index), int(width), int(height), int(P), int(value))
    elif value < 0:
        _capacity_cache.pop((device_index, width, height, P), None)
    else:
        _capacity_cache[(device_index, width, height, P)] = int(value)


def clear_capacity_hints() -> None:
    if _C is not None:
        _C.clear_capacity()
    _capacity_cache.clear()


def set_upstream_scale_mod_grad(on: bool) -> None:
    """Upstream-quirk switch (include/gmsplat.h): with it on, dL/dscale comes back WITHOUT the scale_modifier factor -- what the
    public upstream CUDA backward is believed to return (unverifiable here: the submodule is absent).  Inert at scale_modifier = 1.0,
    which is what every render() of the reference passes.  Also: GMS_UPSTREAM_SCALE_MOD_GRAD=1 in the environment."""
    _lib.load().gms_set_upstream_scale_mod_grad(1 if on else 0)


def set_deterministic(on: bool) -> None:
    """Deterministic-reduction mode (include/gmsplat.h, gms_set_deterministic): the backward passes of the rasterizer and of the
    mesh op sum in a fixed order with no float atomics -- two runs on the same inputs give bit-identical gradients.  Slower;
    process-wide; also switched on by the environment variable GAMES_HIP_DETERMINISTIC=1."""
    _lib.load().gms_set_deterministic(1 if on else 0)


def deterministic() -> bool:
    return bool(_lib.load().gms_get_deterministic())


DEFERRED_OVERFLOW = "GMS_DEFERRED_OVERFLOW"


def set_deferred_counts(on: bool) -> None:
    """Deferred read-back of the frame's instance count (include/gmsplat.h, count_ticket_out).  Off (default): every
    differentiated forward waits for the count inside the call, as the upstream binding's blocking `num_rendered` read-back does, and re-runs
    a frame that outgrew its buffers.  On: the forward only enqueues and the count is read at the START OF THE BACKWARD -- the host thread
    runs ahead of the GPU by the loss and everything up to `backward()`, which is what keeps a slow or shared host from stalling the GPU.
    The price: a frame that outgrew the capacity hint (1.25 x the recent maximum) can only be REPORTED then, its image is already
    incomplete: the backward raises RuntimeError containing `DEFERRED_OVERFLOW` and the caller redoes the step (games_hip.train.training
    and bench.py do; the reference's train.py cannot, hence opt-in).  Frames rendered under no_grad are never deferred.  Also:
    GMS_DEFER_COUNTS=1 in the environment."""
    if _C is None:
        raise NotImplementedError("the deferred read-back needs the _C extension module (GMS_BINDING=ctypes drives the blocking form)")
    _C.set_deferred_counts(bool(on))


def deferred_counts() -> bool:
    return bool(_C is not None and _C.deferred_counts())


def set_sh_factor_mode(on: bool) -> None:
    """Factorised SH gradient for multi-view steps (include/gmsplat.h, gms_sh_grad_expand).  While on, a backward on the SH path
    writes NO dL/dsh (the `shs` gradient is None): it queues a [P+1,3] tensor -- rows 0..P-1 the clamp-masked dL/dcolour of that
    view, row P its camera centre.  Take the queue with `take_sh_factors()`, exchange it between ranks (3 floats per Gaussian
    per view instead of 48) and form the SH gradient with `sh_grad_expand`.  games_hip.ddp.ShFactorExchange does all of it."""
    if _C is None:
        raise NotImplementedError("the factorised SH gradient needs the _C extension module (GMS_BINDING=ctypes has no such path)")
    _C.set_sh_factor_mode(bool(on))


def sh_factor_mode() -> bool:
    return _C is not None and bool(_C.sh_factor_mode())


def take_sh_factors(device=None) -> list:
    """The factors queued by the backward calls ON `device` (a torch.device / index; None = the current device) since the last
    take (or since the mode was switched on), oldest first.  The queue is per device and bounded (256 factors)."""
    if _C is None:
        return []
    idx = -1 if device is None else (device if isinstance(device, int) else (torch.device(device).index if torch.device(device).index is not None else -1))
    return list(_C.take_sh_factors(int(idx)))


def sh_grad_expand(factors: torch.Tensor, means3D: torch.Tensor, sh_degree: int, dL_dsh: torch.Tensor,
                   dL_dsh_rest: Optional[torch.Tensor] = None, accumulate: bool = False) -> None:
    """dL_dsh (+)= sum_v Y(normalize(means3D - campos_v)) (x) factor_v for `factors` = [V,P+1,3] (as queued by the backward
    calls, stacked; possibly gathered from all ranks), views in index order.  dL_dsh is [P,M,3], or the [P,1,3] DC block
    together with dL_dsh_rest = [P,M-1,3] (the split storage of `_features_dc` / `_features_rest`)."""
    if _C is None:
        raise NotImplementedError("sh_grad_expand needs the _C extension module")
    _C.sh_grad_expand(factors, means3D, int(sh_degree), dL_dsh, dL_dsh_rest if dL_dsh_rest is not None else torch.Tensor(), bool(accumulate))


_keep_buffers = False


def keep_buffers(on: bool = True) -> None:
    """Diagnostics: while on, the most recent forward's scratch tensors (and radii) stay referenced, so that `last_stats()`
    can report `visible` / `interactions` and `raw_buffers()` returns them.  Off by default: holding them would keep the
    previous frame's scratch alive while the next forward allocates its own (double the scratch working set)."""
    global _keep_buffers
    _keep_buffers = bool(on)
    if _C is not None:
        _C.set_keep_buffers(bool(on))
    if not on:
        for k in ("radii", "image", "binning", "geom"):
            _last_stats.pop(k, None)


def raw_buffers() -> dict:
    """The scratch tensors of the most recent forward (geom / binning / image byte buffers): diagnostics only; needs
    `keep_buffers(True)` before that forward."""
    d = _C.last_stats() if _C is not None else _last_stats
    return {k: d.get(k) for k in ("geom", "binning", "image")}


def last_stats() -> dict:
    """Counters of the most recent forward call: num_rendered (N), capacity hint used, deepest_tile (instances in the
    deepest tile); after `keep_buffers(True)` also `visible` (Gaussians with radius > 0) and `interactions` (computed
    on request: one device sync)."""
    d = dict(_C.last_stats()) if _C is not None else dict(_last_stats)
    radii = d.pop("radii", None)
    if radii is not None:
        d["visible"] = int((radii > 0).sum())
    d.pop("binning", None)
    d.pop("geom", None)
    image = d.pop("image", None)
    if image is not None and image.numel():
        # sum over pixels of the stored n_contrib.  QUADRANT kernels (GMS_MICRO=0, deep frames): the 1-based position IN THE TILE'S
        # LIST of the last splat the pixel composited = the (pixel, splat) steps of a per-pixel front-to-back walk, SURVEY.md 8(d)
        # "interactions", comparable with the oracle's count.  MICRO-TILE kernels (the default on shallow frames): seg * L + the
        # 1-based position in the pixel's 4x4 BLOCK's pre-filtered list -- the steps of a walk over entries that reach the block,
        # several times smaller and NOT comparable with the figure above or with earlier rounds (`interactions_kind` says which).
        off = int(_lib.load().gms_image_n_contrib_offset(d["width"], d["height"]))
        d["interactions"] = int(image[off:off + 4 * d["width"] * d["height"]].view(torch.int32).sum(dtype=torch.int64))
        micro = bool(d.get("used_micro", int(_lib.load().gms_last_used_micro())))      # the library's own decision for that frame
        d["interactions_kind"] = "micro_tile_block_list_positions" if micro else "tile_list_positions"
    return d


def _f32c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


class _Scratch:
    """Holds the three byte tensors the C side asks for through its resize callbacks."""

    def __init__(self, device):
        self.device = device
        self.tensors = {}
        self.error = None

        def make(name):
            def cb(_ctx, nbytes):
                try:
                    t = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=self.device)
                    self.tensors[name] = t
                    return t.data_ptr()
                except Exception as e:  # noqa: BLE001 - reported through the error code path
                    self.error = e
                    return None
            return _lib.ALLOC_FN(cb)

        self.geom_cb, self.binning_cb, self.image_cb = make("geom"), make("binning"), make("image")


_empties = {}


def _empty(device):
    e = _empties.get(device)
    if e is None:
        e = _empties[device] = torch.empty(0, device=device)
    return e


def _cpu_copy(x):
    if torch.is_tensor(x):
        return x.detach().cpu().clone()
    if isinstance(x, SplitSH):
        return (_cpu_copy(x.dc), _cpu_copy(x.rest))
    if isinstance(x, tuple):
        return tuple(_cpu_copy(v) for v in x)
    return x


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, visible_out=None):
    """`visible_out`: optional bool[P] tensor the preprocess kernel fills with radii > 0 (saves the elementwise pass).
    With `raster_settings.debug` (pipe.debug, arguments/__init__.py:68) every kernel is followed by a sync + error check
    and, as upstream does, a failing forward leaves its inputs in `snapshot_fw.dump` (torch.save of CPU copies)."""
    if raster_settings.debug:
        snapshot = _cpu_copy((means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                              tuple(raster_settings)))
        try:
            return _rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                        raster_settings, visible_out)
        except Exception:
            torch.save(snapshot, "snapshot_fw.dump")
            print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
            raise
    return _rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                raster_settings, visible_out)


def _rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                         raster_settings, visible_out=None):
    if _C is not None:
        rs = raster_settings
        e = _empty(means3D.device)
        rest = e
        if isinstance(sh, SplitSH):
            if sh.dc.shape[1] == 1 and sh.rest.shape[1] == 15 and sh.dc.is_cuda:
                sh, rest = sh.dc, sh.rest
            else:
                sh = sh.full()
        return _C.rasterize(means3D, means2D, sh, rest, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs.bg,
                            rs.viewmatrix, rs.projmatrix, rs.campos, int(rs.image_height), int(rs.image_width), float(rs.tanfovx),
                            float(rs.tanfovy), float(rs.scale_modifier), int(rs.sh_degree), bool(rs.prefiltered),
                            bool(rs.antialiasing), bool(rs.debug), visible_out if visible_out is not None else e,
                            os.environ.get("GMS_SYNC_BINNING", "0") != "1")
    if isinstance(sh, SplitSH):
        if sh.dc.shape[1] == 1 and sh.rest.shape[1] == 15 and sh.dc.is_cuda:
            return _RasterizeGaussians.apply(means3D, means2D, sh.dc, colors_precomp, opacities, scales, rotations,
                                             cov3Ds_precomp, raster_settings, sh.rest, visible_out)
        sh = sh.full()
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, None, visible_out)


_accum_cache = {}      # (device index, stream, P) -> zeroed [P,16] gradient-record buffer


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings, sh_rest=None, visible_out=None):
        rs = raster_settings
        lib = _lib.load()
        _lib.require_gpu(means3D, rs.bg, rs.viewmatrix, rs.projmatrix, rs.campos)
        device = means3D.device
        # an output the loss does not use (inverse depth in train.py) arrives as None in backward instead of a
        # materialised zero image: the backward kernels then skip that channel altogether
        ctx.set_materialize_grads(False)
        if means3D.dim() != 2 or means3D.shape[1] != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")
        P = int(means3D.shape[0])
        H, W = int(rs.image_height), int(rs.image_width)

        means3D = _f32c(means3D)
        sh = _f32c(sh) if sh.numel() else sh
        colors_precomp = _f32c(colors_precomp) if colors_precomp.numel() else colors_precomp
        opacities = _f32c(opacities)
        scales = _f32c(scales) if scales.numel() else scales
        rotations = _f32c(rotations) if rotations.numel() else rotations
        cov3Ds_precomp = _f32c(cov3Ds_precomp) if cov3Ds_precomp.numel() else cov3Ds_precomp
        bg, view, proj, campos = (_f32c(t.to(device)) for t in (rs.bg, rs.viewmatrix, rs.projmatrix, rs.campos))
        if sh.numel() and (sh.dim() != 3 or sh.shape[0] != P or sh.shape[2] != 3):
            raise RuntimeError("sh must have dimensions (num_points, num_coeffs, 3)")
        M = int(sh.shape[1]) if sh.numel() else 0
        if sh_rest is not None:                 # split storage: sh = DC block, sh_rest = the other M-1 coefficients
            sh_rest = _f32c(sh_rest)
            M = int(sh.shape[1] + sh_rest.shape[1])

        color = torch.empty((3, H, W), dtype=torch.float32, device=device)
        invdepth = torch.empty((1, H, W), dtype=torch.float32, device=device)
        radii = torch.empty((P,), dtype=torch.int32, device=device)

        scratch = _Scratch(device)
        key = (device.index, W, H, P)
        hint = 0
        # (deterministic mode: no hint -- segment length / kernel choice then depend on this frame alone, not on earlier ones)
        if os.environ.get("GMS_SYNC_BINNING", "0") != "1" and not lib.gms_get_deterministic():
            prev = _capacity_cache.get(key)
            if prev is not None:
                hint = _quantize_capacity(int(prev * 1.25) + 4096)
        num_units = C.c_int64(0)
        a = _lib.RasterForwardArgs(
            P=P, D=int(rs.sh_degree), M=M, width=W, height=H,
            background=_lib.ptr(bg), means3D=_lib.ptr(means3D), shs=_lib.ptr(sh), shs_rest=_lib.ptr(sh_rest),
            colors_precomp=_lib.ptr(colors_precomp), opacities=_lib.ptr(opacities), scales=_lib.ptr(scales),
            rotations=_lib.ptr(rotations), cov3D_precomp=_lib.ptr(cov3Ds_precomp), viewmatrix=_lib.ptr(view),
            projmatrix=_lib.ptr(proj), campos=_lib.ptr(campos), scale_modifier=float(rs.scale_modifier),
            tan_fovx=float(rs.tanfovx), tan_fovy=float(rs.tanfovy), prefiltered=int(bool(rs.prefiltered)),
            antialiasing=int(bool(rs.antialiasing)), debug=int(bool(rs.debug)),
            out_color=_lib.ptr(color), out_invdepth=_lib.ptr(invdepth), radii=_lib.ptr(radii),
            visible=_lib.ptr(visible_out) if visible_out is not None else None,
            num_units_out=C.addressof(num_units),
            geom_alloc=scratch.geom_cb, geom_ctx=None, binning_alloc=scratch.binning_cb, binning_ctx=None,
            image_alloc=scratch.image_cb, image_ctx=None, binning_capacity_hint=hint)
        with _lib.on_device(device):
            stream = _lib.stream_ptr(device)
            num_rendered = lib.gms_rasterize_forward(C.byref(a), C.c_void_p(stream))
        if num_rendered < 0:
            if scratch.error is not None:
                raise scratch.error
            _lib.check(num_rendered, "gms_rasterize_forward")
        _capacity_cache[key] = max(int(num_rendered), int(0.97 * _capacity_cache.get(key, 0)))
        _last_stats.update(num_rendered=int(num_rendered), num_units=int(num_units.value), capacity_hint=hint, P=P, width=W, height=H,
                           deepest_tile=int(lib.gms_last_deepest_tile()), used_micro=int(lib.gms_last_used_micro()))
        if _keep_buffers:
            _last_stats.update(radii=radii, image=scratch.tensors.get("image"), binning=scratch.tensors.get("binning"),
                               geom=scratch.tensors.get("geom"))

        ctx.raster_settings = rs
        ctx.num_rendered = int(num_rendered)
        ctx.num_units = int(num_units.value)
        ctx.binning_capacity = hint if (hint > 0 and num_rendered <= hint) else max(int(num_rendered), 1)
        ctx.M = M
        empty = torch.empty(0, device=device)
        ctx.split = sh_rest is not None
        ctx.save_for_backward(means3D, sh, sh_rest if sh_rest is not None else empty, colors_precomp, opacities, scales,
                              rotations, cov3Ds_precomp, radii,
                              scratch.tensors.get("geom", empty), scratch.tensors.get("binning", empty),
                              scratch.tensors.get("image", empty), bg, view, proj, campos)
        ctx.mark_non_differentiable(radii)
        return color, radii, invdepth

    @staticmethod
    def backward(ctx, grad_color, _grad_radii, grad_invdepth):
        lib = _lib.load()
        rs = ctx.raster_settings
        (means3D, sh, sh_rest, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, radii, geom, binning, image,
         bg, view, proj, campos) = ctx.saved_tensors
        device = means3D.device
        P = int(means3D.shape[0])
        H, W = int(rs.image_height), int(rs.image_width)
        M = ctx.M
        use_sh, use_cov = sh.numel() > 0, cov3Ds_precomp.numel() > 0

        grad_color = _f32c(grad_color) if grad_color is not None else torch.zeros((3, H, W), device=device)
        grad_invdepth = _f32c(grad_invdepth) if grad_invdepth is not None else None

        # one zeroed 64-byte gradient record per Gaussian for the blend kernel's atomics.  The buffer is kept per
        # (device, stream, P): the backward kernels leave it zero again (grad_accum_rezero), so there is no
        # 64*P-byte memset per iteration.  Calls on one stream are ordered, so sharing it between them is safe.
        stream = _lib.stream_ptr(device)
        accum_key = (device.index, stream, P)
        grad_accum = _accum_cache.pop(accum_key, None)
        if grad_accum is None:
            grad_accum = torch.zeros((max(P, 1), 16), dtype=torch.float32, device=device)
        dL_dmeans2D = torch.empty((P, 3), dtype=torch.float32, device=device)
        dL_dopacity = torch.empty(opacities.shape, dtype=torch.float32, device=device)
        dL_dcolors = torch.empty((P, 3), dtype=torch.float32, device=device) if not use_sh else None
        dL_dmeans3D = torch.empty((P, 3), dtype=torch.float32, device=device)
        dL_dsh = torch.empty(sh.shape, dtype=torch.float32, device=device) if use_sh else None
        dL_dsh_rest = torch.empty(sh_rest.shape, dtype=torch.float32, device=device) if ctx.split else None
        dL_dscales = torch.empty((P, 3), dtype=torch.float32, device=device) if not use_cov else None
        dL_drot = torch.empty((P, 4), dtype=torch.float32, device=device) if not use_cov else None
        dL_dcov3D = torch.empty((P, 6), dtype=torch.float32, device=device) if use_cov else None

        a = _lib.RasterBackwardArgs(
            P=P, D=int(rs.sh_degree), M=M, width=W, height=H, num_rendered=ctx.num_rendered,
            binning_capacity=ctx.binning_capacity,
            background=_lib.ptr(bg), means3D=_lib.ptr(means3D), shs=_lib.ptr(sh),
            shs_rest=_lib.ptr(sh_rest) if ctx.split else None, colors_precomp=_lib.ptr(colors_precomp),
            opacities=_lib.ptr(opacities), scales=_lib.ptr(scales), rotations=_lib.ptr(rotations),
            cov3D_precomp=_lib.ptr(cov3Ds_precomp), viewmatrix=_lib.ptr(view), projmatrix=_lib.ptr(proj),
            campos=_lib.ptr(campos), scale_modifier=float(rs.scale_modifier), tan_fovx=float(rs.tanfovx),
            tan_fovy=float(rs.tanfovy), antialiasing=int(bool(rs.antialiasing)), debug=int(bool(rs.debug)),
            radii=_lib.ptr(radii), geom_buffer=_lib.ptr(geom), binning_buffer=_lib.ptr(binning),
            image_buffer=_lib.ptr(image), dL_dout_color=_lib.ptr(grad_color), dL_dout_invdepth=_lib.ptr(grad_invdepth),
            grad_accum=_lib.ptr(grad_accum), dL_dmeans2D=_lib.ptr(dL_dmeans2D), dL_dopacity=_lib.ptr(dL_dopacity),
            dL_dcolors=_lib.ptr(dL_dcolors), dL_dmeans3D=_lib.ptr(dL_dmeans3D),
            dL_dcov3D=_lib.ptr(dL_dcov3D), dL_dsh=_lib.ptr(dL_dsh), dL_dsh_rest=_lib.ptr(dL_dsh_rest),
            dL_dscales=_lib.ptr(dL_dscales),
            dL_drotations=_lib.ptr(dL_drot), grad_accum_rezero=1, num_units=ctx.num_units)
        if P > 0:
            try:
                with _lib.on_device(device):
                    _lib.check(lib.gms_rasterize_backward(C.byref(a), C.c_void_p(stream)), "gms_rasterize_backward")
            except Exception:
                if rs.debug:        # upstream: the inputs of a failing backward go to snapshot_bw.dump
                    torch.save(_cpu_copy((means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, radii,
                                          grad_color, tuple(rs))), "snapshot_bw.dump")
                    print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise
        # only a call that completed hands its (re-zeroed) buffer back; a failed one lets it go
        if len(_accum_cache) >= 8:
            _accum_cache.clear()
        _accum_cache[accum_key] = grad_accum
        return (dL_dmeans3D, dL_dmeans2D, dL_dsh, dL_dcolors, dL_dopacity, dL_dscales, dL_drot,
                dL_dcov3D, None, dL_dsh_rest, None)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """bool[P]: Gaussian centre passes the near-plane test of the current camera."""
        rs = self.raster_settings
        if _C is not None:
            with torch.no_grad():
                return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)
        lib = _lib.load()
        _lib.require_gpu(positions)
        with torch.no_grad():
            pos = _f32c(positions)
            present = torch.empty((pos.shape[0],), dtype=torch.uint8, device=pos.device)
            view, proj = _f32c(rs.viewmatrix.to(pos.device)), _f32c(rs.projmatrix.to(pos.device))
            with _lib.on_device(pos.device):
                stream = _lib.stream_ptr(pos.device)
                _lib.check(lib.gms_mark_visible(int(pos.shape[0]), _lib.ptr(pos), _lib.ptr(view), _lib.ptr(proj),
                                                _lib.ptr(present), C.c_void_p(stream)), "gms_mark_visible")
        return present.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        e = torch.Tensor([]).to(means3D.device)
        shs = e if shs is None else shs
        colors_precomp = e if colors_precomp is None else colors_precomp
        scales = e if scales is None else scales
        rotations = e if rotations is None else rotations
        cov3D_precomp = e if cov3D_precomp is None else cov3D_precomp
        # `visibility_filter` (radii > 0, renderer/gaussian_renderer/__init__.py:108) comes out of the preprocess kernel;
        # callers that know about it read `rasterizer.visibility_filter` instead of launching the comparison
        vis = torch.empty((means3D.shape[0],), dtype=torch.bool, device=means3D.device) if means3D.is_cuda and means3D.shape[0] else None
        out = rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, rs, vis)
        self.visibility_filter = vis
        return out
