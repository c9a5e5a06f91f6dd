// torch_binding.cpp -- `diff_gaussian_rasterization._C`: the PyTorch-ROCm extension module over the C ABI of
// include/gmsplat.h (libgmsplat.so holds every kernel; nothing here launches one itself).
//
// What it provides
//   * the three entry points of the upstream binding the reference imports (renderer/gaussian_renderer/__init__.py:14
//     -> diff_gaussian_rasterization/__init__.py -> `from . import _C`), with upstream's argument order and return tuples:
//         rasterize_gaussians, rasterize_gaussians_backward, mark_visible
//   * the training fast path: the autograd node itself in C++ (`rasterize`), so one Python call per render reaches the
//     kernels without ctypes marshalling or a Python autograd.Function on the way back (train.py:100-108 is host-bound
//     otherwise: ~0.7 ms of Python per iteration against ~0.7 ms of kernels);
//   * the same for the mesh-face -> Gaussian op (`mesh_to_gaussians`, games/mesh_splatting/scene/gaussian_mesh_model.py:86-169),
//     the fused L1+SSIM loss (`l1_ssim`, train.py:106-107) and the multi-tensor Adam step (`adam_step`, train.py:147).
// torch supplies device memory, the current stream and the autograd graph: plumbing, not arithmetic.  No CPU path: CPU
// tensors raise.
#include <torch/extension.h>

// torch-ROCm presents its devices as type "cuda": the guard / stream accessors are the "masquerading" ones
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <atomic>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "../../include/gmsplat.h"

namespace {

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

inline void *stream_of(const Tensor &t) { return (void *)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }
inline const float *cf(const Tensor &t) { return (t.defined() && t.numel()) ? t.data_ptr<float>() : nullptr; }
inline float *mf(const Tensor &t) { return (t.defined() && t.numel()) ? t.data_ptr<float>() : nullptr; }

inline Tensor f32c(const Tensor &t)
{
    if (!t.defined() || t.numel() == 0) return t;
    Tensor r = t.scalar_type() == torch::kFloat ? t : t.to(torch::kFloat);
    return r.is_contiguous() ? r : r.contiguous();
}

inline void require_gpu(const Tensor &t)
{
    TORCH_CHECK(!t.defined() || t.numel() == 0 || t.is_cuda(),
                "diff_gaussian_rasterization (MI355X/HIP build): tensors must live on a GPU; there is no CPU path in the "
                "product (the CPU oracle lives under oracle/ for tests only)");
}

inline void check_rc(int64_t rc, const char *what)
{
    TORCH_CHECK(rc >= 0, what, " failed (", rc, "): ", gms_last_error());
}

// resize callbacks of the C ABI: the caller's (torch's caching) allocator owns all scratch
struct Slot { Tensor t; c10::Device dev; bool failed; };
void *alloc_cb(void *ctx, size_t bytes)
{
    Slot *s = static_cast<Slot *>(ctx);
    try {
        s->t = torch::empty({(int64_t)(bytes > 0 ? bytes : 1)}, torch::TensorOptions().dtype(torch::kUInt8).device(s->dev));
        return s->t.data_ptr();
    } catch (...) {
        s->failed = true;
        return nullptr;
    }
}

// (device, W, H, P) -> slowly decaying maximum of the instances rendered by recent calls (the binning capacity hint)
std::mutex g_mu;
std::map<std::tuple<int, int, int, int64_t>, int64_t> g_capacity;
// (device, stream, P) -> zeroed [P,16] gradient-record buffer (the backward kernels leave it zero again)
std::map<std::tuple<int, void *, int64_t>, Tensor> g_accum;

// Counters of the most recent forward.  The scratch tensors themselves are referenced only while `keep_buffers` is on
// (diagnostics: last_stats()["interactions"], raw_buffers()): a permanent reference would keep the previous frame's scratch
// alive while the next forward allocates its own, i.e. double the scratch working set of the caching allocator.
// Capacity hints are rounded UP to four significant bits (steps of 6-12 %).  The binning buffer is sized by the hint, and a
// size that creeps up frame by frame -- an animated mesh that grows 0.5 % per frame -- is a fresh hipMalloc in the caching
// allocator on every frame (10 ms against a 1.4 ms render at config-5 size); on the coarse grid the size changes once per
// ~15 such frames and both neighbours stay cached.
static int64_t quantize_capacity(int64_t x)
{
    if (x < 16) return x;
    const int s = 63 - __builtin_clzll((unsigned long long)x) - 3;
    return ((x + ((int64_t)1 << s) - 1) >> s) << s;
}

struct LastCall { int64_t num_rendered = 0, num_units = 0, hint = 0, P = 0; int W = 0, H = 0; Tensor radii, image, binning, geom; } g_last;
std::atomic<bool> g_keep_buffers{false};

struct Forward {
    int64_t num_rendered = 0, num_units = 0, capacity = 0;
    int64_t ticket[2] = {0, 0}, launched_units = 0;      // deferred read-back (set_deferred_counts): the counts are redeemed at the start of the backward
    Tensor color, radii, invdepth, geom, binning, image;
};
// Deferred read-back of the instance count (gmsplat.h, count_ticket_out): opt-in, because an overflowed frame can
// only be REPORTED at the start of its backward (the loss has been computed on an incomplete image by then): the backward raises
// "GMS_DEFERRED_OVERFLOW" and the training loop redoes the step (games_hip/train.py, bench.py); the reference's own loop cannot.
static std::atomic<int> g_defer_counts{-1};
bool deferred_counts()
{
    int v = g_defer_counts.load();
    if (v < 0) { const char *e = getenv("GMS_DEFER_COUNTS"); int expected = -1; g_defer_counts.compare_exchange_strong(expected, (e && atoi(e) != 0) ? 1 : 0); v = g_defer_counts.load(); }
    return v != 0;
}
void set_deferred_counts(bool on) { g_defer_counts.store(on ? 1 : 0); }
// the mesh backward inside preprocess_bwd for frames rendered straight from the mesh (GmsRasterBackwardArgs.mesh, ABI 8); on by default
static std::atomic<int> g_fused_mesh_bwd{-1};
bool fused_mesh_backward()
{
    int v = g_fused_mesh_bwd.load();
    if (v < 0) { const char *e = getenv("GMS_TRAIN_FUSED_BWD"); int expected = -1; g_fused_mesh_bwd.compare_exchange_strong(expected, (e && atoi(e) == 0) ? 0 : 1); v = g_fused_mesh_bwd.load(); }
    return v != 0;
}
void set_fused_mesh_backward(bool on) { g_fused_mesh_bwd.store(on ? 1 : 0); }
// what a frame rendered straight from a mesh stores for its backward (GmsRasterForwardArgs.mesh_out_*, ABI 6)
struct MeshOut { Tensor xyz, scaling_act, rotation_unit, opacity_act; };

Forward forward_core(const Tensor &bg_, const Tensor &means3D_, const Tensor &sh_, const Tensor &sh_rest_, const Tensor &colors_,
                     const Tensor &opac_, const Tensor &scales_, const Tensor &rots_, const Tensor &cov_, const Tensor &view_,
                     const Tensor &proj_, const Tensor &campos_, int64_t H, int64_t W, double tanx, double tany, double mod, int64_t D,
                     bool prefiltered, bool aa, bool debug, const Tensor &visible, bool use_hint, const GmsMeshArgs *mesh = nullptr,
                     const MeshOut *mesh_out = nullptr, bool may_defer = false)
{
    // (`mesh`: the forward-only frame straight from a mesh, gmsplat.h; `means3D_` then only carries the device and P -- the SH DC tensor)
    require_gpu(means3D_); require_gpu(bg_); require_gpu(view_); require_gpu(proj_); require_gpu(campos_);
    TORCH_CHECK(mesh || (means3D_.dim() == 2 && means3D_.size(1) == 3), "means3D must have dimensions (num_points, 3)");
    const auto dev = means3D_.device();
    c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(dev);
    const int64_t P = means3D_.size(0);
    Tensor means3D = f32c(means3D_), sh = f32c(sh_), sh_rest = f32c(sh_rest_), colors = f32c(colors_), opac = f32c(opac_);
    Tensor scales = f32c(scales_), rots = f32c(rots_), cov = f32c(cov_);
    Tensor bg = f32c(bg_.to(dev)), view = f32c(view_.to(dev)), proj = f32c(proj_.to(dev)), campos = f32c(campos_.to(dev));
    const bool has_sh = sh.defined() && sh.numel() > 0;
    if (has_sh) TORCH_CHECK(sh.dim() == 3 && sh.size(0) == P && sh.size(2) == 3, "sh must have dimensions (num_points, num_coeffs, 3)");
    int64_t M = has_sh ? sh.size(1) : 0;
    const bool split = sh_rest.defined() && sh_rest.numel() > 0;
    if (split) M = sh.size(1) + sh_rest.size(1);

    auto fopt = torch::TensorOptions().dtype(torch::kFloat).device(dev);
    Forward f;
    f.color = torch::empty({3, H, W}, fopt);
    f.invdepth = torch::empty({1, H, W}, fopt);
    f.radii = torch::empty({P}, fopt.dtype(torch::kInt));

    int64_t hint = 0;
    const auto key = std::make_tuple((int)dev.index(), (int)W, (int)H, P);
    // deterministic mode (gmsplat.h): no capacity hint -- the segment length and the choice of compositing kernels then depend on the
    // frame alone (the exact instance count), not on what earlier frames of this shape looked like: run 1 == run 2 bit for bit
    if (use_hint && !gms_get_deterministic()) {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_capacity.find(key);
        if (it != g_capacity.end()) hint = quantize_capacity(it->second + it->second / 4 + 4096);
    }
    // Stream capture (torch.cuda.graph): the forward must not touch the host.  gms_rasterize_forward is then asked for its
    // launches-only form; the frame's counts stay on the device (games_hip.animate.GraphedAnimation reads them after a replay).
    hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing((hipStream_t)stream_of(means3D_), &cap_status) == hipSuccess && cap_status != hipStreamCaptureStatusNone;
    if (capturing)
        TORCH_CHECK(hint > 0 && P > 0, "rasterizing inside a stream capture needs the capacity hint of this shape: render it at least once on the "
                                        "same stream before the capture (not in deterministic mode)");
    Slot geom{Tensor(), dev, false}, binning{Tensor(), dev, false}, image{Tensor(), dev, false};
    int64_t num_units = 0;
    GmsRasterForwardArgs a{};
    a.P = (int32_t)P; a.D = (int32_t)D; a.M = (int32_t)M; a.width = (int32_t)W; a.height = (int32_t)H;
    a.background = cf(bg); a.means3D = cf(means3D); a.shs = cf(sh); a.shs_rest = split ? cf(sh_rest) : nullptr;
    a.colors_precomp = cf(colors); a.opacities = cf(opac); a.scales = cf(scales); a.rotations = cf(rots);
    a.cov3D_precomp = cf(cov); a.viewmatrix = cf(view); a.projmatrix = cf(proj); a.campos = cf(campos);
    a.scale_modifier = (float)mod; a.tan_fovx = (float)tanx; a.tan_fovy = (float)tany;
    a.prefiltered = prefiltered; a.antialiasing = aa; a.debug = debug;
    a.out_color = mf(f.color); a.out_invdepth = mf(f.invdepth); a.radii = P ? f.radii.data_ptr<int32_t>() : nullptr;
    a.geom_alloc = alloc_cb; a.geom_ctx = &geom; a.binning_alloc = alloc_cb; a.binning_ctx = &binning;
    a.image_alloc = alloc_cb; a.image_ctx = &image;
    a.binning_capacity_hint = hint;
    a.visible = (visible.defined() && visible.numel()) ? static_cast<uint8_t *>(visible.data_ptr()) : nullptr;
    a.num_units_out = &num_units;
    a.no_host_wait = capturing ? 1 : 0;
    a.mesh = mesh;
    if (mesh) { a.means3D = nullptr; a.opacities = nullptr; a.scales = nullptr; a.rotations = nullptr; }
    if (mesh && mesh_out) {
        a.mesh_out_xyz = mf(mesh_out->xyz); a.mesh_out_scaling_act = mf(mesh_out->scaling_act);
        a.mesh_out_rotation_unit = mf(mesh_out->rotation_unit); a.mesh_out_opacity_act = mf(mesh_out->opacity_act);
    }
    int64_t ticket[2] = {0, 0};
    const bool defer = may_defer && hint > 0 && !capturing && P > 0 && deferred_counts();
    if (defer) a.count_ticket_out = ticket;
    const int64_t n = gms_rasterize_forward(&a, stream_of(means3D));
    TORCH_CHECK(!(geom.failed || binning.failed || image.failed), "scratch allocation failed (out of device memory?)");
    check_rc(n, "gms_rasterize_forward");
    f.num_rendered = n; f.num_units = num_units;
    f.capacity = (hint > 0 && n <= hint) ? hint : (n > 0 ? n : 1);
    f.geom = geom.t; f.binning = binning.t; f.image = image.t;
    if (defer) { f.ticket[0] = ticket[0]; f.ticket[1] = ticket[1]; f.launched_units = gms_last_launched_units(); f.num_rendered = -1; f.capacity = hint; }
    {
        std::lock_guard<std::mutex> lk(g_mu);
        int64_t &c = g_capacity[key];
        if (!capturing && !defer) c = std::max(n, (int64_t)(0.97 * (double)c));      // (a captured / deferred call returns the capacity, not a count)
        g_last.num_rendered = n; g_last.num_units = num_units; g_last.hint = hint; g_last.P = P; g_last.W = (int)W; g_last.H = (int)H;
        if (g_keep_buffers.load()) { g_last.radii = f.radii; g_last.image = f.image; g_last.binning = f.binning; g_last.geom = f.geom; }
    }
    return f;
}

// Start of a backward whose forward deferred its counts: wait for them (they arrived long ago on a GPU-bound loop), check that the frame
// fitted what it was launched for, feed the capacity hint.  Returns {instances, work units}.
std::pair<int64_t, int64_t> redeem_counts(const int64_t *ticket, int64_t capacity, int64_t launched_units, const Tensor &like, int64_t W, int64_t H, int64_t P)
{
    int64_t units = 0, deepest = 0;
    const int64_t n = gms_rasterize_forward_counts(ticket, (int32_t)W, (int32_t)H, (int32_t)P, &units, &deepest, stream_of(like));
    check_rc(n, "gms_rasterize_forward_counts");
    {
        std::lock_guard<std::mutex> lk(g_mu);
        int64_t &c = g_capacity[std::make_tuple((int)like.device().index(), (int)W, (int)H, P)];
        c = std::max(n, (int64_t)(0.97 * (double)c));
        g_last.num_rendered = n; g_last.num_units = units;
    }
    TORCH_CHECK(n <= capacity && units <= launched_units,
                "GMS_DEFERRED_OVERFLOW: the frame held ", n, " instances / ", units, " work units but was launched for ", capacity, " / ", launched_units,
                " (deferred read-back of the instance count: diff_gaussian_rasterization.set_deferred_counts): its image is incomplete -- "
                "redo this step (the capacity hint has been raised; or render it with set_deferred_counts(False))");
    return {n, units};
}

struct Backward { Tensor dmeans2D, dcolors, dopacity, dmeans3D, dcov3D, dsh, dsh_rest, dscales, drots; };
// outputs of the mesh backward when it runs inside preprocess_bwd (GmsRasterBackwardArgs.mesh, ABI 8)
struct MeshGrads { Tensor d_vertices, d_alpha, d_scale, d_opacity; };

// The gradient outputs of a backward call.  The autograd fast path allocates them in FORWARD, before the C call (there the
// host runs ahead of the GPU and then waits for the instance count anyway), so that between "N has arrived" and "blend_bwd is
// enqueued" -- the stretch in which a slow host lets the GPU run dry -- no allocation remains.
// ---- factorised SH gradient (multi-view steps; gms_sh_grad_expand in include/gmsplat.h).  While the mode is on, a backward on
// the SH path writes no dL/dsh: it leaves a [P+1,3] tensor -- rows 0..P-1 the clamp-masked dL/dcolour of that view, row P the
// view's camera centre -- in a per-DEVICE list that the caller takes (take_sh_factors), exchanges between ranks and expands.  The
// queue is keyed by device so two models on two GPUs of one process do not take each other's factors, and bounded: a caller that
// switches the mode on and never takes the factors gets an error instead of an ever-growing list of [P+1,3] tensors.
static std::atomic<bool> g_sh_factor{false};
static std::map<int, std::vector<Tensor>> g_factors;
constexpr size_t MAX_QUEUED_FACTORS = 256;

Backward alloc_backward(const Tensor &means3D, const Tensor &opac, const Tensor &sh, const Tensor &sh_rest, const Tensor &cov)
{
    const int64_t P = means3D.size(0);
    auto fopt = torch::TensorOptions().dtype(torch::kFloat).device(means3D.device());
    const bool has_sh = sh.defined() && sh.numel() > 0, split = sh_rest.defined() && sh_rest.numel() > 0;
    const bool has_cov = cov.defined() && cov.numel() > 0;
    Backward b;
    b.dmeans2D = torch::empty({P, 3}, fopt);
    b.dopacity = torch::empty(opac.sizes(), fopt);
    b.dmeans3D = torch::empty({P, 3}, fopt);
    const bool factor = has_sh && g_sh_factor.load();
    if (!has_sh) b.dcolors = torch::empty({P, 3}, fopt);
    if (factor) b.dcolors = torch::empty({P + 1, 3}, fopt);
    if (has_sh && !factor) b.dsh = torch::empty(sh.sizes(), fopt);
    if (split && !factor) b.dsh_rest = torch::empty(sh_rest.sizes(), fopt);
    if (has_cov) b.dcov3D = torch::empty({P, 6}, fopt);
    else { b.dscales = torch::empty({P, 3}, fopt); b.drots = torch::empty({P, 4}, fopt); }
    return b;
}

Backward backward_core(const Tensor &bg, const Tensor &means3D, const Tensor &radii, const Tensor &colors, const Tensor &opac,
                       const Tensor &scales, const Tensor &rots, double mod, const Tensor &cov, const Tensor &view, const Tensor &proj,
                       double tanx, double tany, const Tensor &dL_dcolor_, const Tensor &dL_dinvd_, const Tensor &sh, const Tensor &sh_rest,
                       int64_t D, const Tensor &campos, const Tensor &geom, int64_t R, int64_t capacity, int64_t num_units,
                       const Tensor &binning, const Tensor &image, bool aa, bool debug, const Backward *prealloc = nullptr,
                       const GmsMeshArgs *mesh = nullptr, const MeshGrads *mesh_grads = nullptr)
{
    const auto dev = means3D.device();
    c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(dev);
    const int64_t P = means3D.size(0);
    const bool has_sh = sh.defined() && sh.numel() > 0, split = sh_rest.defined() && sh_rest.numel() > 0;
    const bool has_cov = cov.defined() && cov.numel() > 0;
    int64_t M = has_sh ? sh.size(1) : 0;
    if (split) M = sh.size(1) + sh_rest.size(1);
    const int64_t H = dL_dcolor_.defined() ? dL_dcolor_.size(-2) : 0, W = dL_dcolor_.defined() ? dL_dcolor_.size(-1) : 0;
    auto fopt = torch::TensorOptions().dtype(torch::kFloat).device(dev);
    Tensor gcol = f32c(dL_dcolor_), ginv = f32c(dL_dinvd_);
    void *stream = stream_of(means3D);
    Tensor accum;
    const auto akey = std::make_tuple((int)dev.index(), stream, P);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_accum.find(akey);
        if (it != g_accum.end()) { accum = it->second; g_accum.erase(it); }
    }
    if (!accum.defined()) accum = torch::zeros({std::max<int64_t>(P, 1), 16}, fopt);
    Backward b = prealloc ? *prealloc : alloc_backward(means3D, opac, sh, sh_rest, cov);
    GmsRasterBackwardArgs a{};
    a.P = (int32_t)P; a.D = (int32_t)D; a.M = (int32_t)M; a.width = (int32_t)W; a.height = (int32_t)H;
    a.num_rendered = R; a.binning_capacity = capacity;
    a.background = cf(bg); a.means3D = cf(means3D); a.shs = cf(sh); a.shs_rest = split ? cf(sh_rest) : nullptr;
    a.colors_precomp = cf(colors); a.opacities = cf(opac); a.scales = cf(scales); a.rotations = cf(rots); a.cov3D_precomp = cf(cov);
    a.viewmatrix = cf(view); a.projmatrix = cf(proj); a.campos = cf(campos);
    a.scale_modifier = (float)mod; a.tan_fovx = (float)tanx; a.tan_fovy = (float)tany; a.antialiasing = aa; a.debug = debug;
    a.radii = P ? radii.data_ptr<int32_t>() : nullptr;
    a.geom_buffer = geom.numel() ? geom.data_ptr() : nullptr;
    a.binning_buffer = binning.numel() ? binning.data_ptr() : nullptr;
    a.image_buffer = image.numel() ? image.data_ptr() : nullptr;
    a.dL_dout_color = cf(gcol); a.dL_dout_invdepth = cf(ginv);
    a.grad_accum = mf(accum); a.dL_dmeans2D = mf(b.dmeans2D); a.dL_dopacity = mf(b.dopacity); a.dL_dcolors = mf(b.dcolors);
    a.dL_dmeans3D = mf(b.dmeans3D); a.dL_dcov3D = mf(b.dcov3D); a.dL_dsh = mf(b.dsh); a.dL_dsh_rest = mf(b.dsh_rest);
    a.dL_dscales = mf(b.dscales); a.dL_drotations = mf(b.drots);
    a.grad_accum_rezero = 1; a.num_units = num_units;
    if (mesh && mesh_grads) {
        a.mesh = mesh; a.mesh_dL_dvertices = mf(mesh_grads->d_vertices); a.mesh_dL_dalpha = mf(mesh_grads->d_alpha);
        a.mesh_dL_dscale = mf(mesh_grads->d_scale); a.mesh_dL_d_opacity = mf(mesh_grads->d_opacity);
    }
    a.sh_factor_mode = (has_sh && b.dcolors.defined()) ? 1 : 0;
    a.factor_campos_row = (a.sh_factor_mode && b.dcolors.size(0) == P + 1) ? 1 : 0;      // row P = the camera centre
    if (a.sh_factor_mode) {
        std::lock_guard<std::mutex> lk(g_mu);
        TORCH_CHECK(g_factors[(int)dev.index()].size() < MAX_QUEUED_FACTORS, "factorised SH mode: ", MAX_QUEUED_FACTORS,
                    " factors queued on this device and never taken (call take_sh_factors() every step, or set_sh_factor_mode(False))");
    }
    if (P > 0) check_rc(gms_rasterize_backward(&a, stream), "gms_rasterize_backward");
    if (a.sh_factor_mode) {          // factorised mode: queue the factor (rows 0..P-1 + camera centre) for the exchange
        std::lock_guard<std::mutex> lk(g_mu);
        g_factors[(int)dev.index()].push_back(b.dcolors);
    }
    {   // only a call that completed hands its (re-zeroed) buffer back
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_accum.size() >= 8) g_accum.clear();
        g_accum[akey] = accum;
    }
    return b;
}

// ---------------------------------------------------------------------------------------------- upstream entry points
std::tuple<int64_t, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor>
rasterize_gaussians(const Tensor &background, const Tensor &means3D, const Tensor &colors, const Tensor &opacity, const Tensor &scales,
                    const Tensor &rotations, double scale_modifier, const Tensor &cov3D_precomp, const Tensor &viewmatrix,
                    const Tensor &projmatrix, double tan_fovx, double tan_fovy, int64_t image_height, int64_t image_width,
                    const Tensor &sh, int64_t degree, const Tensor &campos, bool prefiltered, bool antialiasing, bool debug)
{
    // synchronous sizing (no capacity hint): the binning buffer is laid out for exactly `rendered` instances, which is all
    // the upstream-shaped backward call below knows about it
    Forward f = forward_core(background, means3D, sh, Tensor(), colors, opacity, scales, rotations, cov3D_precomp, viewmatrix,
                             projmatrix, campos, image_height, image_width, tan_fovx, tan_fovy, scale_modifier, degree, prefiltered,
                             antialiasing, debug, Tensor(), false);
    return std::make_tuple(f.num_rendered, f.color, f.radii, f.geom, f.binning, f.image, f.invdepth);
}

std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor>
rasterize_gaussians_backward(const Tensor &background, const Tensor &means3D, const Tensor &radii, const Tensor &colors,
                             const Tensor &opacities, const Tensor &scales, const Tensor &rotations, double scale_modifier,
                             const Tensor &cov3D_precomp, const Tensor &viewmatrix, const Tensor &projmatrix, double tan_fovx,
                             double tan_fovy, const Tensor &dL_dout_color, const Tensor &dL_dout_invdepth, const Tensor &sh,
                             int64_t degree, const Tensor &campos, const Tensor &geomBuffer, int64_t R, const Tensor &binningBuffer,
                             const Tensor &imageBuffer, bool antialiasing, bool debug)
{
    Backward b = backward_core(f32c(background), f32c(means3D), radii, f32c(colors), f32c(opacities), f32c(scales), f32c(rotations),
                               scale_modifier, f32c(cov3D_precomp), f32c(viewmatrix), f32c(projmatrix), tan_fovx, tan_fovy, dL_dout_color,
                               dL_dout_invdepth, f32c(sh), Tensor(), degree, f32c(campos), geomBuffer, R, R > 0 ? R : 1, 0, binningBuffer,
                               imageBuffer, antialiasing, debug);
    return std::make_tuple(b.dmeans2D, b.dcolors, b.dopacity, b.dmeans3D, b.dcov3D, b.dsh, b.dscales, b.drots);
}

Tensor mark_visible(const Tensor &means3D, const Tensor &viewmatrix, const Tensor &projmatrix)
{
    require_gpu(means3D);
    c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(means3D.device());
    Tensor pos = f32c(means3D.detach()), view = f32c(viewmatrix.to(means3D.device())), proj = f32c(projmatrix.to(means3D.device()));
    Tensor present = torch::empty({pos.size(0)}, torch::TensorOptions().dtype(torch::kUInt8).device(pos.device()));
    check_rc(gms_mark_visible((int32_t)pos.size(0), cf(pos), cf(view), cf(proj), pos.size(0) ? present.data_ptr<uint8_t>() : nullptr,
                              stream_of(pos)), "gms_mark_visible");
    return present.to(torch::kBool);
}

// ---------------------------------------------------------------------------------------------- autograd fast path
class RasterizeFn : public torch::autograd::Function<RasterizeFn> {
public:
    static variable_list forward(AutogradContext *ctx, Tensor means3D, Tensor means2D, Tensor sh, Tensor sh_rest, Tensor colors,
                                 Tensor opacities, Tensor scales, Tensor rotations, Tensor cov3D, Tensor bg, Tensor view, Tensor proj,
                                 Tensor campos, int64_t H, int64_t W, double tanx, double tany, double mod, int64_t D, bool prefiltered,
                                 bool aa, bool debug, Tensor visible_out, bool use_hint, bool will_backward)
    {
        ctx->set_materialize_grads(false);      // an unused output (inverse depth in train.py) arrives undefined: its channel is skipped
        if (will_backward && means3D.is_cuda()) {
            c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(means3D.device());
            Backward pre = alloc_backward(means3D, opacities, sh, sh_rest, cov3D);
            const Tensor *ts[9] = {&pre.dmeans2D, &pre.dcolors, &pre.dopacity, &pre.dmeans3D, &pre.dcov3D, &pre.dsh, &pre.dsh_rest,
                                   &pre.dscales, &pre.drots};
            for (int k = 0; k < 9; k++)
                if (ts[k]->defined()) ctx->saved_data[std::string("pre") + char('0' + k)] = *ts[k];
            ctx->saved_data["pre"] = true;
        }
        Forward f = forward_core(bg, means3D, sh, sh_rest, colors, opacities, scales, rotations, cov3D, view, proj, campos, H, W, tanx,
                                 tany, mod, D, prefiltered, aa, debug, visible_out, use_hint, nullptr, nullptr, will_backward && use_hint);
        ctx->saved_data["ticket0"] = f.ticket[0]; ctx->saved_data["ticket1"] = f.ticket[1]; ctx->saved_data["launched"] = f.launched_units;
        const auto dev = means3D.device();
        ctx->save_for_backward({f32c(means3D), f32c(sh), f32c(sh_rest), f32c(colors), f32c(opacities), f32c(scales), f32c(rotations),
                                f32c(cov3D), f.radii, f.geom, f.binning, f.image, f32c(bg.to(dev)), f32c(view.to(dev)),
                                f32c(proj.to(dev)), f32c(campos.to(dev))});
        ctx->saved_data["R"] = f.num_rendered; ctx->saved_data["units"] = f.num_units; ctx->saved_data["cap"] = f.capacity;
        ctx->saved_data["tanx"] = tanx; ctx->saved_data["tany"] = tany; ctx->saved_data["mod"] = mod; ctx->saved_data["D"] = D;
        ctx->saved_data["aa"] = aa; ctx->saved_data["debug"] = debug; ctx->saved_data["H"] = H; ctx->saved_data["W"] = W;
        ctx->mark_non_differentiable({f.radii});
        return {f.color, f.radii, f.invdepth};
    }

    static variable_list backward(AutogradContext *ctx, variable_list grads)
    {
        auto s = ctx->get_saved_variables();
        const Tensor &means3D = s[0], &sh = s[1], &sh_rest = s[2], &colors = s[3], &opac = s[4], &scales = s[5], &rots = s[6], &cov = s[7];
        const Tensor &radii = s[8], &geom = s[9], &binning = s[10], &image = s[11], &bg = s[12], &view = s[13], &proj = s[14], &campos = s[15];
        const int64_t H = ctx->saved_data["H"].toInt(), W = ctx->saved_data["W"].toInt();
        Tensor gcol = grads[0].defined() ? grads[0] : torch::zeros({3, H, W}, means3D.options());
        Backward pre;
        const bool have_pre = ctx->saved_data.count("pre") > 0;
        if (have_pre) {        // handed out once: a second backward through a retained graph allocates its own outputs
            Tensor *ts[9] = {&pre.dmeans2D, &pre.dcolors, &pre.dopacity, &pre.dmeans3D, &pre.dcov3D, &pre.dsh, &pre.dsh_rest, &pre.dscales,
                             &pre.drots};
            for (int k = 0; k < 9; k++) {
                const std::string key = std::string("pre") + char('0' + k);
                if (ctx->saved_data.count(key)) { *ts[k] = ctx->saved_data[key].toTensor(); ctx->saved_data.erase(key); }
            }
            ctx->saved_data.erase("pre");
        }
        int64_t R = ctx->saved_data["R"].toInt(), units = ctx->saved_data["units"].toInt();
        if (ctx->saved_data["ticket0"].toInt() != 0) {
            c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(means3D.device());
            const int64_t ticket[2] = {ctx->saved_data["ticket0"].toInt(), ctx->saved_data["ticket1"].toInt()};
            std::tie(R, units) = redeem_counts(ticket, ctx->saved_data["cap"].toInt(), ctx->saved_data["launched"].toInt(), means3D, W, H, means3D.size(0));
            ctx->saved_data["ticket0"] = (int64_t)0; ctx->saved_data["R"] = R; ctx->saved_data["units"] = units;      // (a second backward through a retained graph)
        }
        Backward b = backward_core(bg, means3D, radii, colors, opac, scales, rots, ctx->saved_data["mod"].toDouble(), cov, view, proj,
                                   ctx->saved_data["tanx"].toDouble(), ctx->saved_data["tany"].toDouble(), gcol, grads[2], sh, sh_rest,
                                   ctx->saved_data["D"].toInt(), campos, geom, R, ctx->saved_data["cap"].toInt(),
                                   units, binning, image, ctx->saved_data["aa"].toBool(),
                                   ctx->saved_data["debug"].toBool(), have_pre ? &pre : nullptr);
        Tensor none;
        if (sh.defined() && sh.numel() > 0) b.dcolors = none;          // (factorised mode: the factor is not a gradient of `colors`)
        return {b.dmeans3D, b.dmeans2D, b.dsh, b.dsh_rest, b.dcolors, b.dopacity, b.dscales, b.drots, b.dcov3D,
                none, none, none, none, none, none, none, none, none, none, none, none, none, none, none, none};
    }
};

std::tuple<Tensor, Tensor, Tensor> rasterize(const Tensor &means3D, const Tensor &means2D, const Tensor &sh, const Tensor &sh_rest,
                                             const Tensor &colors, const Tensor &opacities, const Tensor &scales, const Tensor &rotations,
                                             const Tensor &cov3D, const Tensor &bg, const Tensor &view, const Tensor &proj,
                                             const Tensor &campos, int64_t H, int64_t W, double tanx, double tany, double mod, int64_t D,
                                             bool prefiltered, bool aa, bool debug, const Tensor &visible_out, bool use_hint)
{
    const bool will_backward = at::GradMode::is_enabled() &&
        (means3D.requires_grad() || means2D.requires_grad() || sh.requires_grad() || sh_rest.requires_grad() || colors.requires_grad() ||
         opacities.requires_grad() || scales.requires_grad() || rotations.requires_grad() || cov3D.requires_grad());
    auto out = RasterizeFn::apply(means3D, means2D, sh, sh_rest, colors, opacities, scales, rotations, cov3D, bg, view, proj, campos, H,
                                  W, tanx, tany, mod, D, prefiltered, aa, debug, visible_out, use_hint, will_backward);
    return std::make_tuple(out[0], out[1], out[2]);
}

void set_sh_factor_mode(bool on)
{
    g_sh_factor = on;
    std::lock_guard<std::mutex> lk(g_mu);
    g_factors.clear();
}
bool sh_factor_mode() { return g_sh_factor.load(); }
// device < 0: the current device
std::vector<Tensor> take_sh_factors(int64_t device)
{
    if (device < 0) { int d = 0; if (hipGetDevice(&d) != hipSuccess) d = 0; device = d; }
    std::lock_guard<std::mutex> lk(g_mu);
    std::vector<Tensor> out;
    auto it = g_factors.find((int)device);
    if (it != g_factors.end()) { out.swap(it->second); g_factors.erase(it); }
    return out;
}

// dsh (+)= sum_v Y(dir_v) (x) factor_v;  factors = [V,P+1,3] as queued by the backward calls (possibly gathered from all ranks)
void sh_grad_expand(const Tensor &factors, const Tensor &means3D, int64_t D, Tensor dsh, Tensor dsh_rest, bool accumulate)
{
    require_gpu(means3D);
    c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(means3D.device());
    TORCH_CHECK(factors.dim() == 3 && factors.size(2) == 3 && factors.size(1) == means3D.size(0) + 1 && factors.is_contiguous() &&
                factors.scalar_type() == torch::kFloat && factors.device() == means3D.device(),
                "sh_grad_expand: factors must be a contiguous float32 [V, P+1, 3] tensor on the device of means3D");
    TORCH_CHECK(dsh.defined() && dsh.is_contiguous() && dsh.scalar_type() == torch::kFloat, "sh_grad_expand: dsh must be contiguous float32");
    const int64_t P = means3D.size(0), V = factors.size(0);
    const bool split = dsh_rest.defined() && dsh_rest.numel() > 0;
    if (split) TORCH_CHECK(dsh_rest.is_contiguous() && dsh_rest.scalar_type() == torch::kFloat, "sh_grad_expand: dsh_rest must be contiguous float32");
    const int64_t M = split ? dsh.numel() / std::max<int64_t>(P * 3, 1) + dsh_rest.numel() / std::max<int64_t>(P * 3, 1) : dsh.numel() / std::max<int64_t>(P * 3, 1);
    Tensor pos = f32c(means3D.detach());
    Tensor campos = factors.select(1, P).contiguous();        // [V,3]
    GmsShGradExpandArgs a{};
    a.P = (int32_t)P; a.D = (int32_t)D; a.M = (int32_t)M; a.V = (int32_t)V; a.means3D = cf(pos); a.campos = cf(campos);
    a.factors = cf(factors); a.factor_stride = (P + 1) * 3; a.dL_dsh = mf(dsh); a.dL_dsh_rest = split ? mf(dsh_rest) : nullptr;
    a.accumulate = accumulate; a.debug = 0;
    if (P > 0 && V > 0) check_rc(gms_sh_grad_expand(&a, stream_of(means3D)), "gms_sh_grad_expand");
}

void set_keep_buffers(bool on)
{
    g_keep_buffers = on;
    if (!on) { std::lock_guard<std::mutex> lk(g_mu); g_last.radii = Tensor(); g_last.image = Tensor(); g_last.binning = Tensor(); g_last.geom = Tensor(); }
}
// drop the cached (all-zero between calls) gradient-record buffers, e.g. after a fault-injection test dirtied one
void clear_accum() { std::lock_guard<std::mutex> lk(g_mu); g_accum.clear(); }

py::dict last_stats()
{
    std::lock_guard<std::mutex> lk(g_mu);
    py::dict d;
    d["num_rendered"] = g_last.num_rendered; d["num_units"] = g_last.num_units; d["capacity_hint"] = g_last.hint;
    d["P"] = g_last.P; d["width"] = g_last.W; d["height"] = g_last.H; d["deepest_tile"] = gms_last_deepest_tile();
    d["used_micro"] = (int)gms_last_used_micro();
    if (g_last.radii.defined()) { d["radii"] = g_last.radii; d["image"] = g_last.image; d["binning"] = g_last.binning; d["geom"] = g_last.geom; }
    return d;
}

void set_capacity(int64_t device, int64_t W, int64_t H, int64_t P, int64_t value)
{
    std::lock_guard<std::mutex> lk(g_mu);
    const auto key = std::make_tuple((int)device, (int)W, (int)H, P);
    if (value < 0) g_capacity.erase(key); else g_capacity[key] = value;
}
void clear_capacity() { std::lock_guard<std::mutex> lk(g_mu); g_capacity.clear(); }

// ---------------------------------------------------------------------------------------------- mesh-face -> Gaussian
GmsMeshArgs mesh_args(const Tensor &vertices, const Tensor &faces, const Tensor &_alpha, const Tensor &_scale, int64_t mode, int64_t spf,
                      const Tensor &fso, const Tensor &sf, bool fused, const Tensor &_opacity)
{
    GmsMeshArgs a{};
    a.F = (int32_t)faces.size(0); a.V = (int32_t)vertices.size(0); a.P = _scale.numel(); a.splats_per_face = (int32_t)spf;
    a.alpha_mode = (int32_t)mode; a.vertices = cf(vertices); a.faces = faces.numel() ? faces.data_ptr<int64_t>() : nullptr;
    a.face_splat_offset = (fso.defined() && fso.numel()) ? fso.data_ptr<int32_t>() : nullptr;
    a.splat_face = (sf.defined() && sf.numel()) ? sf.data_ptr<int32_t>() : nullptr;
    a._alpha = cf(_alpha); a._scale = cf(_scale); a.fused_activations = fused; a._opacity = cf(_opacity);
    return a;
}

// Forward-only frame of the animated render drivers (games_hip.animate): mesh -> image in the rasterizer's own launches, K0 inside
// the preprocess thread (GmsRasterForwardArgs.mesh).  Returns (image, radii, inverse depth, radii > 0).
std::tuple<Tensor, Tensor, Tensor, Tensor> render_mesh_forward(const Tensor &vertices, const Tensor &faces, const Tensor &_alpha, const Tensor &_scale,
                                                       const Tensor &_opacity, int64_t mode, int64_t spf, const Tensor &splat_face,
                                                       const Tensor &sh_dc, const Tensor &sh_rest, const Tensor &bg, const Tensor &view,
                                                       const Tensor &proj, const Tensor &campos, int64_t H, int64_t W, double tanx, double tany,
                                                       double mod, bool aa, bool debug)
{
    require_gpu(vertices); require_gpu(_alpha); require_gpu(_scale); require_gpu(_opacity); require_gpu(sh_dc); require_gpu(sh_rest);
    TORCH_CHECK(faces.scalar_type() == torch::kInt64 && faces.is_contiguous() && faces.is_cuda(), "faces must be a contiguous int64 device tensor");
    Tensor v = f32c(vertices), al = f32c(_alpha), sc = f32c(_scale), op = f32c(_opacity);
    const int64_t P = sc.numel();
    TORCH_CHECK(al.numel() == 3 * P && op.numel() == P && sh_dc.size(0) == P && sh_rest.size(0) == P, "render_mesh_forward: P mismatch");
    TORCH_CHECK(faces.dim() == 2 && faces.size(1) == 3, "faces must have dimensions (num_faces, 3)");
    TORCH_CHECK(v.dim() == 2 && v.size(1) == 3, "vertices must have dimensions (num_vertices, 3)");
    TORCH_CHECK(faces.device() == sh_dc.device() && v.device() == sh_dc.device(), "render_mesh_forward: mesh and SH tensors live on different devices");
    if (spf > 0) { TORCH_CHECK(faces.size(0) * spf == P, "render_mesh_forward: ", faces.size(0), " faces x ", spf, " splats per face != ", P, " Gaussians"); }
    else { TORCH_CHECK(splat_face.defined() && splat_face.numel() == P, "render_mesh_forward: non-uniform splat counts need splat_face [P]"); }
    TORCH_CHECK(sh_dc.dim() == 3 && sh_dc.size(1) == 1 && sh_rest.dim() == 3 && sh_rest.size(1) == 15, "render_mesh_forward needs split degree-3 SH storage ([P,1,3] + [P,15,3])");
    GmsMeshArgs m = mesh_args(v, faces, al, sc, mode, spf, Tensor(), splat_face, true, op);
    // (P stands in for means3D: forward_core reads the device and the count from its first tensor argument)
    Tensor stand_in = sh_dc.view({P, 3});
    Tensor visible = torch::empty({P}, sh_dc.options().dtype(torch::kBool));          // radii > 0, written by the preprocess kernel
    Forward f = forward_core(bg, stand_in, sh_dc, sh_rest, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), view, proj, campos, H, W, tanx, tany, mod,
                             3, false, aa, debug, visible, true, &m);
    return std::make_tuple(f.color, f.radii, f.invdepth, visible);
}

class MeshFn : public torch::autograd::Function<MeshFn> {
public:
    static variable_list forward(AutogradContext *ctx, Tensor vertices_, Tensor faces_, Tensor alpha_, Tensor scale_, int64_t mode,
                                 int64_t spf, Tensor fso_, Tensor sf_, bool fused, Tensor opacity_, bool vertex_grad)
    {
        require_gpu(vertices_); require_gpu(faces_); require_gpu(alpha_); require_gpu(scale_);
        ctx->set_materialize_grads(false);
        const auto dev = vertices_.device();
        c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(dev);
        const bool has_op = opacity_.defined() && opacity_.numel() > 0;
        TORCH_CHECK(!has_op || fused, "_opacity fusion needs fused_activations=True");
        Tensor vertices = f32c(vertices_), _alpha = f32c(alpha_), _scale = f32c(scale_), _opacity = f32c(opacity_);
        Tensor faces = faces_.scalar_type() == torch::kLong ? faces_.contiguous() : faces_.to(torch::kLong).contiguous();
        Tensor fso = fso_.defined() && fso_.numel() ? fso_.to(torch::kInt).contiguous() : Tensor();
        Tensor sf = sf_.defined() && sf_.numel() ? sf_.to(torch::kInt).contiguous() : Tensor();
        const int64_t P = _scale.numel();
        auto fopt = torch::TensorOptions().dtype(torch::kFloat).device(dev);
        Tensor alpha = torch::empty_like(_alpha), xyz = torch::empty({P, 3}, fopt), scaling = torch::empty({P, 3}, fopt);
        Tensor rotation = torch::empty({P, 4}, fopt), sact, runit, oact;
        if (fused) { sact = torch::empty({P, 3}, fopt); runit = torch::empty({P, 4}, fopt); }
        if (has_op) oact = torch::empty_like(_opacity);
        // the vertex-gradient buffer of the coming backward is cleared by spare blocks of this launch
        Tensor vgrad;
        if (vertex_grad) vgrad = torch::empty_like(vertices);
        GmsMeshArgs a = mesh_args(vertices, faces, _alpha, _scale, mode, spf, fso, sf, fused, _opacity);
        a.prezero = mf(vgrad); a.prezero_count = vgrad.defined() ? vgrad.numel() : 0;
        check_rc(gms_mesh_to_gaussians_forward(&a, mf(alpha), mf(xyz), mf(scaling), mf(rotation), mf(sact), mf(runit), mf(oact),
                                               stream_of(vertices)), "gms_mesh_to_gaussians_forward");
        ctx->save_for_backward({vertices, faces, _alpha, _scale, fso.defined() ? fso : torch::empty({0}, fopt),
                                sf.defined() ? sf : torch::empty({0}, fopt), has_op ? _opacity : torch::empty({0}, fopt),
                                vgrad.defined() ? vgrad : torch::empty({0}, fopt)});
        ctx->saved_data["mode"] = mode; ctx->saved_data["spf"] = spf; ctx->saved_data["fused"] = fused; ctx->saved_data["used"] = false;
        if (fused) {
            ctx->mark_non_differentiable({alpha, scaling, rotation});
            if (has_op) return {alpha, xyz, scaling, rotation, sact, runit, oact};
            return {alpha, xyz, scaling, rotation, sact, runit};
        }
        ctx->mark_non_differentiable({alpha});
        return {alpha, xyz, scaling, rotation};
    }

    static variable_list backward(AutogradContext *ctx, variable_list g)
    {
        auto s = ctx->get_saved_variables();
        const Tensor &vertices = s[0], &faces = s[1], &_alpha = s[2], &_scale = s[3];
        Tensor fso = s[4].numel() ? s[4] : Tensor(), sf = s[5].numel() ? s[5] : Tensor(), _opacity = s[6].numel() ? s[6] : Tensor();
        const bool fused = ctx->saved_data["fused"].toBool();
        const auto dev = vertices.device();
        c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(dev);
        const int64_t P = _scale.numel();
        auto fopt = torch::TensorOptions().dtype(torch::kFloat).device(dev);
        Tensor g_xyz = g[1], g_scaling = fused ? (g.size() > 4 ? g[4] : Tensor()) : g[2], g_rot = fused ? (g.size() > 5 ? g[5] : Tensor()) : g[3];
        Tensor g_op = (fused && g.size() > 6) ? g[6] : Tensor();
        auto gz = [&](const Tensor &t, int64_t c) { return t.defined() ? f32c(t) : torch::zeros({P, c}, fopt); };
        g_xyz = gz(g_xyz, 3); g_scaling = gz(g_scaling, 3); g_rot = gz(g_rot, 4);
        // the pre-cleared vertex-gradient buffer serves ONE backward; a second one through a retained graph clears its own
        Tensor d_vertices;
        bool prezeroed = false;
        if (s[7].numel() && !ctx->saved_data["used"].toBool()) { d_vertices = s[7]; prezeroed = true; ctx->saved_data["used"] = true; }
        else d_vertices = torch::empty_like(vertices);
        Tensor d_alpha = torch::empty_like(_alpha), d_scale = torch::empty_like(_scale), d_opacity;
        const bool want_op = _opacity.defined() && g_op.defined();
        if (want_op) { g_op = f32c(g_op); d_opacity = torch::empty_like(_opacity); }
        GmsMeshArgs a = mesh_args(vertices, faces, _alpha, _scale, ctx->saved_data["mode"].toInt(), ctx->saved_data["spf"].toInt(), fso, sf,
                                  fused, _opacity);
        a.vertex_grad_prezeroed = prezeroed;
        check_rc(gms_mesh_to_gaussians_backward(&a, cf(g_xyz), cf(g_scaling), cf(g_rot), want_op ? cf(g_op) : nullptr, mf(d_vertices),
                                                mf(d_alpha), mf(d_scale), want_op ? mf(d_opacity) : nullptr, stream_of(vertices)),
                 "gms_mesh_to_gaussians_backward");
        Tensor none;
        return {d_vertices, none, d_alpha, d_scale, none, none, none, none, none, d_opacity, none};
    }
};

std::vector<Tensor> mesh_to_gaussians(const Tensor &vertices, const Tensor &faces, const Tensor &_alpha, const Tensor &_scale, int64_t mode,
                                      int64_t spf, const Tensor &fso, const Tensor &sf, bool fused, const Tensor &_opacity)
{
    // (whether a backward will follow is decided here: inside forward() the graph node may not exist)
    return MeshFn::apply(vertices, faces, _alpha, _scale, mode, spf, fso, sf, fused, _opacity,
                         at::GradMode::is_enabled() && vertices.requires_grad());
}

// ---------------------------------------------------------------------------------------------- training frame straight from the mesh
// train.py:100-108 with the K0 launch of train.py:154-157 folded into the rasterizer's preprocess thread (GmsRasterForwardArgs.mesh +
// mesh_out_*, ABI 6): ONE autograd node from (vertices, _alpha, _scale, _opacity, SH) to the image.  Forward: no K0 launch -- the
// preprocess thread derives its Gaussian from the face and stores xyz / activated scale / unit quaternion / sigmoid opacity (44 bytes
// per Gaussian instead of K0's 84 written + 44 read back).  Backward: gms_rasterize_backward on those four tensors, then
// gms_mesh_to_gaussians_backward on its gradients -- the same two kernels groups as the two-node graph, one node's worth of host work.
class RenderMeshFn : public torch::autograd::Function<RenderMeshFn> {
public:
    static variable_list forward(AutogradContext *ctx, Tensor vertices_, Tensor faces, Tensor alpha_, Tensor scale_, Tensor opacity_,
                                 Tensor sh_dc, Tensor sh_rest, Tensor means2D, int64_t mode, int64_t spf, Tensor splat_face, Tensor bg,
                                 Tensor view, Tensor proj, Tensor campos, int64_t H, int64_t W, double tanx, double tany, double mod,
                                 bool aa, bool debug, bool vertex_grad, bool will_backward, int64_t sh_degree)
    {
        ctx->set_materialize_grads(false);
        TORCH_CHECK(sh_degree >= 0 && sh_degree <= 3, "render_mesh: active SH degree ", sh_degree, " (storage is degree 3: 0 .. 3)");
        require_gpu(vertices_); require_gpu(alpha_); require_gpu(scale_); require_gpu(opacity_); require_gpu(sh_dc); require_gpu(sh_rest);
        TORCH_CHECK(faces.scalar_type() == torch::kInt64 && faces.is_contiguous() && faces.is_cuda(), "faces must be a contiguous int64 device tensor");
        const auto dev = vertices_.device();
        c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(dev);
        Tensor v = f32c(vertices_), al = f32c(alpha_), sc = f32c(scale_), op = f32c(opacity_), dc = f32c(sh_dc), rest = f32c(sh_rest);
        const int64_t P = sc.numel();
        TORCH_CHECK(al.numel() == 3 * P && op.numel() == P && dc.size(0) == P && rest.size(0) == P, "render_mesh: P mismatch");
        TORCH_CHECK(dc.dim() == 3 && dc.size(1) == 1 && rest.dim() == 3 && rest.size(1) == 15, "render_mesh needs split degree-3 SH storage ([P,1,3] + [P,15,3])");
        TORCH_CHECK(faces.dim() == 2 && faces.size(1) == 3, "faces must have dimensions (num_faces, 3)");
        TORCH_CHECK(v.dim() == 2 && v.size(1) == 3, "vertices must have dimensions (num_vertices, 3)");
        if (spf > 0) { TORCH_CHECK(faces.size(0) * spf == P, "render_mesh: ", faces.size(0), " faces x ", spf, " splats per face != ", P, " Gaussians"); }
        else { TORCH_CHECK(splat_face.defined() && splat_face.numel() == P, "render_mesh: non-uniform splat counts need splat_face [P]"); }
        auto fopt = torch::TensorOptions().dtype(torch::kFloat).device(dev);
        MeshOut mo{torch::empty({P, 3}, fopt), torch::empty({P, 3}, fopt), torch::empty({P, 4}, fopt), torch::empty_like(op)};
        Tensor vgrad;
        if (will_backward && vertex_grad) vgrad = torch::empty_like(v);
        if (will_backward) {          // the backward's outputs, allocated while the host is ahead of the GPU (see RasterizeFn)
            Backward pre = alloc_backward(mo.xyz, mo.opacity_act, dc, rest, Tensor());
            const Tensor *ts[9] = {&pre.dmeans2D, &pre.dcolors, &pre.dopacity, &pre.dmeans3D, &pre.dcov3D, &pre.dsh, &pre.dsh_rest, &pre.dscales, &pre.drots};
            for (int k = 0; k < 9; k++)
                if (ts[k]->defined()) ctx->saved_data[std::string("pre") + char('0' + k)] = *ts[k];
            ctx->saved_data["pre"] = true;
        }
        Tensor sf = (splat_face.defined() && splat_face.numel()) ? splat_face.to(torch::kInt).contiguous() : Tensor();
        GmsMeshArgs m = mesh_args(v, faces, al, sc, mode, spf, Tensor(), sf, true, op);
        m.prezero = mf(vgrad); m.prezero_count = vgrad.defined() ? vgrad.numel() : 0;
        Tensor stand_in = dc.view({P, 3});          // (forward_core reads the device and the count from its first tensor argument)
        // `visibility_filter` (radii > 0, renderer/gaussian_renderer/__init__.py:108) out of the preprocess kernel, as on the two-node route
        Tensor visible = torch::empty({P}, fopt.dtype(torch::kBool));
        Forward f = forward_core(bg, stand_in, dc, rest, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), view, proj, campos, H, W, tanx, tany, mod,
                                 sh_degree, false, aa, debug, visible, true, &m, &mo, will_backward);
        ctx->saved_data["D"] = sh_degree;
        ctx->saved_data["ticket0"] = f.ticket[0]; ctx->saved_data["ticket1"] = f.ticket[1]; ctx->saved_data["launched"] = f.launched_units;
        ctx->save_for_backward({v, faces, al, sc, op, sf.defined() ? sf : torch::empty({0}, fopt), vgrad.defined() ? vgrad : torch::empty({0}, fopt),
                                mo.xyz, mo.scaling_act, mo.rotation_unit, mo.opacity_act, dc, rest, f.radii, f.geom, f.binning, f.image,
                                f32c(bg.to(dev)), f32c(view.to(dev)), f32c(proj.to(dev)), f32c(campos.to(dev))});
        ctx->saved_data["R"] = f.num_rendered; ctx->saved_data["units"] = f.num_units; ctx->saved_data["cap"] = f.capacity;
        ctx->saved_data["tanx"] = tanx; ctx->saved_data["tany"] = tany; ctx->saved_data["mod"] = mod; ctx->saved_data["aa"] = aa;
        ctx->saved_data["debug"] = debug; ctx->saved_data["H"] = H; ctx->saved_data["W"] = W; ctx->saved_data["mode"] = mode;
        ctx->saved_data["spf"] = spf; ctx->saved_data["used"] = false;
        // the stored Gaussians are by-products for the model's attributes: gradients reach the mesh parameters through THIS node
        ctx->mark_non_differentiable({f.radii, mo.xyz, mo.scaling_act, mo.rotation_unit, mo.opacity_act, visible});
        return {f.color, f.radii, f.invdepth, mo.xyz, mo.scaling_act, mo.rotation_unit, mo.opacity_act, visible};
    }

    static variable_list backward(AutogradContext *ctx, variable_list grads)
    {
        auto s = ctx->get_saved_variables();
        const Tensor &v = s[0], &faces = s[1], &al = s[2], &sc = s[3], &op = s[4];
        Tensor sf = s[5].numel() ? s[5] : Tensor();
        const Tensor &xyz = s[7], &sact = s[8], &runit = s[9], &oact = s[10], &dc = s[11], &rest = s[12];
        const Tensor &radii = s[13], &geom = s[14], &binning = s[15], &image = s[16], &bg = s[17], &view = s[18], &proj = s[19], &campos = s[20];
        const auto dev = v.device();
        c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(dev);
        const int64_t H = ctx->saved_data["H"].toInt(), W = ctx->saved_data["W"].toInt();
        Tensor gcol = grads[0].defined() ? grads[0] : torch::zeros({3, H, W}, xyz.options());
        Backward pre;
        const bool have_pre = ctx->saved_data.count("pre") > 0;
        if (have_pre) {
            Tensor *ts[9] = {&pre.dmeans2D, &pre.dcolors, &pre.dopacity, &pre.dmeans3D, &pre.dcov3D, &pre.dsh, &pre.dsh_rest, &pre.dscales, &pre.drots};
            for (int k = 0; k < 9; k++) {
                const std::string key = std::string("pre") + char('0' + k);
                if (ctx->saved_data.count(key)) { *ts[k] = ctx->saved_data[key].toTensor(); ctx->saved_data.erase(key); }
            }
            ctx->saved_data.erase("pre");
        }
        int64_t R = ctx->saved_data["R"].toInt(), units = ctx->saved_data["units"].toInt();
        if (ctx->saved_data["ticket0"].toInt() != 0) {
            const int64_t ticket[2] = {ctx->saved_data["ticket0"].toInt(), ctx->saved_data["ticket1"].toInt()};
            std::tie(R, units) = redeem_counts(ticket, ctx->saved_data["cap"].toInt(), ctx->saved_data["launched"].toInt(), xyz, W, H, xyz.size(0));
            ctx->saved_data["ticket0"] = (int64_t)0; ctx->saved_data["R"] = R; ctx->saved_data["units"] = units;
        }
        Tensor d_vertices;
        bool prezeroed = false;
        if (s[6].numel() && !ctx->saved_data["used"].toBool()) { d_vertices = s[6]; prezeroed = true; ctx->saved_data["used"] = true; }
        else d_vertices = torch::empty_like(v);
        Tensor d_alpha = torch::empty_like(al), d_scale = torch::empty_like(sc), d_opacity = torch::empty_like(op);
        GmsMeshArgs a = mesh_args(v, faces, al, sc, ctx->saved_data["mode"].toInt(), ctx->saved_data["spf"].toInt(), Tensor(), sf, true, op);
        a.vertex_grad_prezeroed = prezeroed;
        // The mesh backward INSIDE preprocess_bwd (ABI 8): the thread of a Gaussian carries its gradients on through the face -> Gaussian
        // parameterization from registers -- no dL/dxyz / dL/dscale / dL/drot / dL/dopacity tensors, no mesh_bwd launch.  Needs the vertex
        // gradient buffer the forward cleared, 1-4 splats per face, float-atomics mode.  GMS_TRAIN_FUSED_BWD=0 keeps the two launches.
        const bool fused_bwd = fused_mesh_backward() && prezeroed && a.splats_per_face > 0 && a.splats_per_face <= 4 && !gms_get_deterministic() && !g_sh_factor.load();
        MeshGrads mg{d_vertices, d_alpha, d_scale, d_opacity};
        Backward b = backward_core(bg, xyz, radii, Tensor(), oact, sact, runit, ctx->saved_data["mod"].toDouble(), Tensor(), view, proj,
                                   ctx->saved_data["tanx"].toDouble(), ctx->saved_data["tany"].toDouble(), gcol, grads[2], dc, rest,
                                   ctx->saved_data["D"].toInt(), campos, geom,
                                   R, ctx->saved_data["cap"].toInt(), units, binning, image,
                                   ctx->saved_data["aa"].toBool(), ctx->saved_data["debug"].toBool(), have_pre ? &pre : nullptr,
                                   fused_bwd ? &a : nullptr, fused_bwd ? &mg : nullptr);
        if (fused_bwd) {
            Tensor none;
            return {d_vertices, none, d_alpha, d_scale, d_opacity, b.dsh, b.dsh_rest, b.dmeans2D,
                    none, none, none, none, none, none, none, none, none, none, none, none, none, none, none, none, none};
        }
        // ... else through the mesh -> Gaussian parameterization as a launch of its own (fused activations: gradients w.r.t. exp / normalize / sigmoid outputs)
        if (a.splats_per_face <= 0) {
            // (non-uniform splat counts: the per-face part of the mesh backward walks CSR offsets this node does not carry)
            TORCH_CHECK(false, "render_mesh backward: non-uniform splat counts take the two-node graph (mesh_to_gaussians + rasterize)");
        }
        check_rc(gms_mesh_to_gaussians_backward(&a, cf(b.dmeans3D), cf(b.dscales), cf(b.drots), cf(b.dopacity), mf(d_vertices), mf(d_alpha),
                                                mf(d_scale), mf(d_opacity), stream_of(v)), "gms_mesh_to_gaussians_backward");
        Tensor none;
        return {d_vertices, none, d_alpha, d_scale, d_opacity, b.dsh, b.dsh_rest, b.dmeans2D,
                none, none, none, none, none, none, none, none, none, none, none, none, none, none, none, none, none};
    }
};

std::vector<Tensor> render_mesh(const Tensor &vertices, const Tensor &faces, const Tensor &_alpha, const Tensor &_scale, const Tensor &_opacity,
                                const Tensor &sh_dc, const Tensor &sh_rest, const Tensor &means2D, int64_t mode, int64_t spf, const Tensor &splat_face,
                                const Tensor &bg, const Tensor &view, const Tensor &proj, const Tensor &campos, int64_t H, int64_t W, double tanx,
                                double tany, double mod, bool aa, bool debug, int64_t sh_degree)
{
    const bool will_backward = at::GradMode::is_enabled() &&
        (vertices.requires_grad() || _alpha.requires_grad() || _scale.requires_grad() || _opacity.requires_grad() || sh_dc.requires_grad() ||
         sh_rest.requires_grad() || means2D.requires_grad());
    return RenderMeshFn::apply(vertices, faces, _alpha, _scale, _opacity, sh_dc, sh_rest, means2D, mode, spf, splat_face, bg, view, proj, campos, H, W,
                               tanx, tany, mod, aa, debug, vertices.requires_grad(), will_backward, sh_degree);
}

// ---------------------------------------------------------------------------------------------- fused L1 + SSIM
class L1SsimFn : public torch::autograd::Function<L1SsimFn> {
public:
    // Returns {value (0-dim, differentiable), (l1, ssim) (reported, not trained on)}: two views of one 3-float buffer.  A single [3] output
    // that the caller indexes costs the backward a zeros(3) and a copy (SelectBackward) before this node is even reached.
    static variable_list forward(AutogradContext *ctx, Tensor img, Tensor gt, double w_l1, double w_ssim, double bias, bool need_grad)
    {
        require_gpu(img); require_gpu(gt);
        ctx->set_materialize_grads(false);      // (the by-products' gradient would otherwise arrive as a zeros tensor: a fill launch per step)
        TORCH_CHECK(img.sizes() == gt.sizes(), "image shapes differ");
        TORCH_CHECK(img.dim() >= 2, "images must be [..., H, W]");
        c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(img.device());
        Tensor x = f32c(img.detach()), y = f32c(gt.detach());
        const int64_t h = x.size(-2), w = x.size(-1), planes = x.numel() / (h * w);
        auto fopt = x.options();
        Tensor out = torch::empty({3}, fopt);
        Tensor partials = torch::empty({(int64_t)gms_l1_ssim_partials((int32_t)planes, (int32_t)h, (int32_t)w)}, fopt);
        Tensor dmaps;
        if (need_grad) {
            std::vector<int64_t> shp{3};
            for (auto d : x.sizes()) shp.push_back(d);
            dmaps = torch::empty(shp, fopt);
        }
        GmsLossArgs a{(int32_t)planes, (int32_t)h, (int32_t)w, cf(x), cf(y), (float)w_l1, (float)w_ssim, (float)bias};
        check_rc(gms_l1_ssim_forward(&a, mf(dmaps), mf(partials), mf(out), stream_of(x)), "gms_l1_ssim_forward");
        ctx->save_for_backward({x, y, dmaps.defined() ? dmaps : torch::empty({0}, fopt)});
        ctx->saved_data["w_l1"] = w_l1; ctx->saved_data["w_ssim"] = w_ssim; ctx->saved_data["bias"] = bias;
        Tensor value = out.select(0, 0), stats = out.narrow(0, 1, 2);
        ctx->mark_non_differentiable({stats});
        return {value, stats};
    }

    static variable_list backward(AutogradContext *ctx, variable_list g)
    {
        auto s = ctx->get_saved_variables();
        const Tensor &x = s[0], &y = s[1], &dmaps = s[2];
        Tensor none;
        if (!g[0].defined()) return {none, none, none, none, none, none};          // (the value was not used)
        TORCH_CHECK(dmaps.numel() > 0, "l1_ssim backward called but the forward ran without requires_grad");
        c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(x.device());
        const int64_t h = x.size(-2), w = x.size(-1), planes = x.numel() / (h * w);
        // only element 0 (the value) is differentiable; the l1 / ssim by-products are reported, not trained on
        Tensor gv = f32c(g[0].reshape({1}));
        Tensor d_img = torch::empty_like(x);
        GmsLossArgs a{(int32_t)planes, (int32_t)h, (int32_t)w, cf(x), cf(y), (float)ctx->saved_data["w_l1"].toDouble(),
                      (float)ctx->saved_data["w_ssim"].toDouble(), (float)ctx->saved_data["bias"].toDouble()};
        check_rc(gms_l1_ssim_backward(&a, cf(dmaps), cf(gv), mf(d_img), stream_of(x)), "gms_l1_ssim_backward");
        return {d_img, none, none, none, none, none};
    }
};

std::vector<Tensor> l1_ssim(const Tensor &img, const Tensor &gt, double w_l1, double w_ssim, double bias)
{
    return L1SsimFn::apply(img, gt, w_l1, w_ssim, bias, at::GradMode::is_enabled() && img.requires_grad());
}

// ---------------------------------------------------------------------------------------------- multi-tensor Adam
void adam_step(const std::vector<Tensor> &params, const std::vector<Tensor> &grads, const std::vector<Tensor> &exp_avg,
               const std::vector<Tensor> &exp_avg_sq, const std::vector<double> &lrs, const std::vector<int64_t> &steps, double beta1,
               double beta2, double eps)
{
    const size_t n = params.size();
    TORCH_CHECK(grads.size() == n && exp_avg.size() == n && exp_avg_sq.size() == n && lrs.size() == n && steps.size() == n, "adam_step: list sizes differ");
    if (n == 0) return;
    std::vector<GmsAdamTensor> t(n);
    std::vector<Tensor> keep;
    for (size_t i = 0; i < n; i++) {
        require_gpu(params[i]);
        TORCH_CHECK(params[i].scalar_type() == torch::kFloat && params[i].is_contiguous(), "FusedAdam needs contiguous float32 parameters");
        TORCH_CHECK(exp_avg[i].is_contiguous() && exp_avg_sq[i].is_contiguous(), "FusedAdam needs contiguous optimizer state");
        Tensor g = f32c(grads[i]);
        keep.push_back(g);
        t[i] = GmsAdamTensor{params[i].data_ptr<float>(), g.data_ptr<float>(), exp_avg[i].data_ptr<float>(), exp_avg_sq[i].data_ptr<float>(),
                             params[i].numel(), (float)lrs[i], (int32_t)steps[i]};
    }
    c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(params[0].device());
    check_rc(gms_adam_step(t.data(), (int32_t)n, beta1, beta2, eps, stream_of(params[0])), "gms_adam_step");
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "MI355X-native diff_gaussian_rasterization._C (PyTorch-ROCm binding of libgmsplat.so)";
    // The GIL is released for the whole of every call that reaches the kernels ("Threading / streams"): the
    // forward polls the pinned read-back slot for the instance count (~one step of GPU time) and must not hold other Python
    // threads (data loaders, a second stream's renderer) meanwhile.  Nothing below touches a Python object: tensors are
    // at::Tensor handles, allocation goes through the ATen caching allocator, the autograd graph is C++.
    using nogil = py::call_guard<py::gil_scoped_release>;
    m.def("rasterize_gaussians", &rasterize_gaussians, nogil());
    m.def("rasterize_gaussians_backward", &rasterize_gaussians_backward, nogil());
    m.def("mark_visible", &mark_visible, nogil());
    m.def("rasterize", &rasterize, "differentiable rasterization (autograd node in C++)", nogil());
    m.def("mesh_to_gaussians", &mesh_to_gaussians, "differentiable mesh-face -> Gaussian parameterization", nogil());
    m.def("render_mesh_forward", &render_mesh_forward, "forward-only frame straight from a mesh (K0 inside the preprocess thread)", nogil());
    m.def("render_mesh", &render_mesh, "differentiable frame straight from a mesh: [image, radii, invdepth, xyz, scaling_act, rotation_unit, opacity_act]", nogil());
    m.def("l1_ssim", &l1_ssim, "differentiable w_l1 * L1 + w_ssim * SSIM + bias; returns (value [0-dim], [l1, ssim])", nogil());
    m.def("adam_step", &adam_step, nogil());
    m.def("set_keep_buffers", &set_keep_buffers, "keep references to the last forward's scratch tensors (diagnostics only)");
    m.def("clear_accum", &clear_accum);
    m.def("set_sh_factor_mode", &set_sh_factor_mode, "factorised SH gradient: backward calls queue [P+1,3] factors instead of writing dL/dsh");
    m.def("sh_factor_mode", &sh_factor_mode);
    m.def("take_sh_factors", &take_sh_factors, py::arg("device") = -1);
    m.def("sh_grad_expand", &sh_grad_expand, "dsh (+)= sum_v Y(dir_v) (x) factor_v over the [V,P+1,3] factors", nogil());
    m.def("last_stats", &last_stats);
    m.def("set_deferred_counts", &set_deferred_counts, "read the frame's instance count back at the start of the backward instead of inside the forward (opt-in)");
    m.def("deferred_counts", &deferred_counts);
    m.def("set_fused_mesh_backward", &set_fused_mesh_backward, "frames rendered straight from a mesh: run the mesh backward inside preprocess_bwd (default on)");
    m.def("fused_mesh_backward", &fused_mesh_backward);
    m.def("set_capacity", &set_capacity);
    m.def("clear_capacity", &clear_capacity);
    m.def("abi_version", []() { return (int64_t)gms_abi_version(); });
    m.def("image_counts_offset", [](int64_t w, int64_t h) { return (int64_t)gms_image_counts_offset((int32_t)w, (int32_t)h); });
    m.def("last_launched_units", []() { return (int64_t)gms_last_launched_units(); });
}
