/*
 * gmsplat.h -- C ABI of the MI355X-native differentiable Gaussian rasterizer and the fused
 * mesh-face -> Gaussian parameterization (libgmsplat.so, built from
 * gaussian-mesh-splatting_amd/csrc by hipcc --offload-arch=gfx950).
 *
 * Plain pointers and sizes only: no torch types, no C++ in the signatures.  Every pointer
 * marked "device" is an address in the HBM of the GPU that `stream` belongs to.  `stream`
 * is a hipStream_t passed as void* (NULL = the null stream).  The calling thread must have
 * that GPU current (hipSetDevice).  All entry points are re-entrant.  State that outlives a
 * call is per host thread and keyed by (device, stream[, W, H, P]): the always-zero tile
 * counter buffer, the launch-size hints learnt from earlier frames of the same shape, and one
 * pinned 64-byte read-back slot per thread; nothing is shared between threads, devices or
 * streams.
 *
 * What each entry point replaces in the reference (waczjoan/gaussian-mesh-splatting):
 *
 *   gms_rasterize_forward   <- diff_gaussian_rasterization._C.rasterize_gaussians, i.e. what
 *                              `GaussianRasterizer.forward` reaches from
 *                              renderer/gaussian_renderer/__init__.py:94-102 (and the three
 *                              sibling renderers: gaussian_animated_renderer/__init__.py:104-112,
 *                              flame_gaussian_renderer/__init__.py:99-107,
 *                              gaussian_points_animated_renderer/__init__.py:97-105)
 *   gms_rasterize_backward  <- diff_gaussian_rasterization._C.rasterize_gaussians_backward,
 *                              triggered by loss.backward() at train.py:108
 *   gms_mark_visible        <- diff_gaussian_rasterization._C.mark_visible
 *                              (GaussianRasterizer.markVisible; no call site in this tree)
 *   gms_mesh_to_gaussians_forward / _backward
 *                           <- GaussianMeshModel.update_alpha + _calc_xyz + prepare_scaling_rot
 *                              (games/mesh_splatting/scene/gaussian_mesh_model.py:86-169), its
 *                              autograd backward, rot_to_quat_batch (utils/general_utils.py:43-96),
 *                              the multi-mesh loop (games/multi_mesh_splatting/scene/
 *                              gaussian_multi_mesh_model.py:99-199) and the softmax variant
 *                              (games/flame_splatting/scene/gaussian_flame_model.py:195)
 *
 * Matrix layout: exactly what scene/cameras.py:54-56 hands over -- the transposed 4x4, so the
 * mathematical element (row r, col c) is m[4*c + r].  Quaternions are (w,x,y,z) and are used
 * as given (the reference normalises in python, scene/gaussian_model.py:100-101).
 */
#ifndef GMSPLAT_H
#define GMSPLAT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GMS_ABI_VERSION 8   /* 3: GmsRasterBackwardArgs gained factor_campos_row + sh_factor_mode (explicit mode flag), GMS_K_COUNT 17;
                             4: GmsRasterForwardArgs gained no_host_wait (stream-capturable forward), gms_image_counts_offset;
                             5: GmsRasterForwardArgs.mesh (forward-only frame straight from a mesh);
                             6: GmsRasterForwardArgs.mesh_out_* (the fused frame exports what the backward needs: training frames too);
                             7: GmsRasterForwardArgs.count_ticket_out + gms_rasterize_forward_counts (the instance count is read back at the
                                START OF THE BACKWARD instead of inside the forward);
                             8: GmsRasterBackwardArgs.mesh + mesh_dL_* (the mesh backward inside preprocess_bwd for frames rendered from a mesh) */

/* error codes (negative return values) */
#define GMS_OK 0
#define GMS_ERR_INVALID_ARGUMENT (-1)
#define GMS_ERR_ALLOC (-2)       /* a resize callback returned NULL */
#define GMS_ERR_HIP (-3)         /* a HIP runtime call failed; see gms_last_error() */
#define GMS_ERR_CAPACITY (-4)    /* caller-provided capacity too small (retry with the returned need) */

/* Resize callback: must return a device pointer to at least `bytes` bytes, 256-byte aligned,
 * that stays valid until the matching backward has run.  Mirrors the three
 * std::function<char*(size_t)> buffers of the upstream binding (geometry / binning / image
 * state) so the caller's caching allocator owns all scratch. */
typedef void *(*gms_alloc_fn)(void *ctx, size_t bytes);

struct GmsMeshArgs;
typedef struct GmsRasterForwardArgs {
    /* sizes */
    int32_t P;          /* number of Gaussians */
    int32_t D;          /* active SH degree (0..3) */
    int32_t M;          /* SH coefficients stored per Gaussian (16 for degree-3 storage); 0 if no SH */
    int32_t width, height;
    /* inputs (device) */
    const float *background;      /* [3] */
    const float *means3D;         /* [P,3] */
    const float *shs;             /* [P,M,3] or NULL */
    const float *shs_rest;        /* optional split storage (the reference keeps _features_dc [P,1,3] and
                                     _features_rest [P,M-1,3] as separate parameters and concatenates them every
                                     iteration, scene/gaussian_model.py:107-111): when non-NULL, `shs` is the DC
                                     block [P,1,3] and `shs_rest` the remaining [P,M-1,3]; M stays the total */
    const float *colors_precomp;  /* [P,3] or NULL  (exactly one of shs / colors_precomp) */
    const float *opacities;       /* [P] */
    const float *scales;          /* [P,3] or NULL */
    const float *rotations;       /* [P,4] or NULL */
    const float *cov3D_precomp;   /* [P,6] or NULL  (exactly one of (scales,rotations) / cov3D_precomp) */
    const float *viewmatrix;      /* [16] */
    const float *projmatrix;      /* [16] */
    const float *campos;          /* [3] */
    float scale_modifier;
    float tan_fovx, tan_fovy;
    int32_t prefiltered;          /* accepted for API parity; culled points are simply culled */
    int32_t antialiasing;
    int32_t debug;                /* sync + check after every kernel */
    /* outputs (device) */
    float *out_color;             /* [3,H,W] */
    float *out_invdepth;          /* [1,H,W] */
    int32_t *radii;               /* [P] */
    /* scratch owned by the caller through resize callbacks */
    gms_alloc_fn geom_alloc;    void *geom_ctx;
    gms_alloc_fn binning_alloc; void *binning_ctx;
    gms_alloc_fn image_alloc;   void *image_ctx;
    /* Optional: if > 0 the binning buffer is requested up-front for this many (Gaussian,tile)
     * instances and the whole pipeline is enqueued before the host looks at the real count
     * (no pipeline bubble); on overflow the tail of the pipeline is re-run after a resize. */
    int64_t binning_capacity_hint;
    /* Optional output [P] bytes: 1 where radii > 0 -- the `visibility_filter` of renderer/gaussian_renderer/__init__.py:108
     * written by the preprocess kernel instead of a separate elementwise pass.  NULL = not wanted. */
    uint8_t *visible;
    /* Optional HOST pointer: receives the number of (tile, segment) work units of this frame; pass it back to
     * gms_rasterize_backward (num_units) so its launch is sized exactly.  NULL = not wanted. */
    int64_t *num_units_out;
    /* 1: the call only ENQUEUES -- no host wait for the instance count, no overflow re-run, nothing but kernel launches (and one
     * host-visible store from a kernel) on `stream` -- so that it can be captured into a hipGraph (the
     * animated render loops of scripts/render_time_animated.py:68-87 replayed without host work).  Needs binning_capacity_hint > 0
     * and every library-owned buffer of this (thread, device, stream, shape) in place, i.e. at least one ordinary call on the same
     * stream first.  The return value is then the CAPACITY (an upper bound of the instance count unless the frame overflowed); the
     * frame's true counts stay on the device: four uint32 {instances, deepest tile, work units, segment length} at byte offset
     * gms_image_counts_offset() of the image scratch buffer.  A frame with more instances than the capacity, or more work units than
     * the launch was sized for, is INCOMPLETE and the caller must detect it from those counts (games_hip.animate.GraphedAnimation
     * does).  Pass the returned value as num_rendered and binning_capacity, and num_units = 0, to a backward call. */
    int32_t no_host_wait;
    /* Optional (ABI 5): a forward-only frame rendered straight from a mesh -- the animated render loops of
     * scripts/render_time_animated.py:68-87 / scripts/render_flame.py:29-60, where every frame moves the vertices and nothing is
     * differentiated.  When non-NULL the preprocess thread derives its Gaussian from the mesh (barycentric centre, face frame ->
     * activated scale and unit quaternion, sigmoid opacity: the arithmetic of gms_mesh_to_gaussians_forward with fused_activations,
     * bit for bit) and `means3D`, `opacities`, `scales`, `rotations` are ignored (may be NULL): the K0 launch and the 84 bytes per
     * Gaussian it writes disappear.  Needs mesh->P == P, mesh->_opacity, split degree-3 SH STORAGE (shs + shs_rest, M = 16; the ACTIVE degree D is 0 .. 3)
     * and no precomputed colours / covariances.  `mesh->prezero` / `prezero_count` are honoured as in gms_mesh_to_gaussians_forward (the
     * [V,3] buffer the mesh backward will accumulate into is cleared by this launch). */
    const struct GmsMeshArgs *mesh;
    /* ABI 6 -- TRAINING frames straight from the mesh (train.py:100-108 with the K0 launch of train.py:154-157 folded into the
     * preprocess thread).  All four NULL: a forward-only frame, which cannot be handed to gms_rasterize_backward (nothing holds its
     * Gaussians).  All four set: the preprocess thread also stores what it derived -- exactly the xyz / scaling_activated /
     * rotation_unit / opacity_activated outputs of gms_mesh_to_gaussians_forward, bit for bit -- 44 bytes per Gaussian instead of K0's
     * 84 + the 44 this kernel would read back.  The caller then runs gms_rasterize_backward with those four tensors as means3D / scales
     * / rotations / opacities and feeds its gradients to gms_mesh_to_gaussians_backward (fused_activations = 1). */
    float *mesh_out_xyz;           /* [P,3] */
    float *mesh_out_scaling_act;   /* [P,3] */
    float *mesh_out_rotation_unit; /* [P,4] */
    float *mesh_out_opacity_act;   /* [P]   */
    /* ABI 7 -- DEFERRED read-back of the frame's counts (replaces the blocking `num_rendered` read-back the upstream binding does inside its
     * forward).  Optional HOST pointer; non-NULL + binning_capacity_hint > 0 + no_host_wait
     * == 0: the call enqueues the whole pipeline, has the device publish this frame's counts to one of sixteen pinned slots of the calling
     * host thread, stores a TWO-WORD ticket here ({slot, sequence number}) and RETURNS WITHOUT WAITING (return value = the capacity, as with
     * no_host_wait).  The host can then run ahead of the GPU by the loss and everything else up to the backward.  Before the backward the
     * caller redeems the ticket with gms_rasterize_forward_counts() -- from ANY host thread: torch runs backward passes on its own thread --
     * which waits for the counts (long since there on a GPU-bound loop); the caller must compare them with what the frame was launched for --
     * instances <= capacity, work units <= gms_last_launched_units() taken right after the forward -- because nothing re-runs an overflowed
     * frame in this form: its image is incomplete and its backward must not be trusted.  At most sixteen tickets of a forward thread may be
     * outstanding. */
    int64_t *count_ticket_out;      /* -> int64_t[2] */
} GmsRasterForwardArgs;

/* Returns the number of (Gaussian, tile) instances rendered (>= 0) or a negative error code. */
int64_t gms_rasterize_forward(const GmsRasterForwardArgs *args, void *stream);
/* Redeem the two-word ticket of a deferred forward (GmsRasterForwardArgs.count_ticket_out): waits until the device has published the frame's
 * counts, returns the number of instances (>= 0) or a negative error code (an expired ticket: more than sixteen were outstanding), stores
 * the frame's work units / deepest tile, and leaves the launch-size hints of this (device, stream, width, height, P) for the next forward of
 * that shape, as a blocking forward would have.  Any host thread. */
int64_t gms_rasterize_forward_counts(const int64_t *ticket, int32_t width, int32_t height, int32_t P, int64_t *num_units_out,
                                     int64_t *deepest_tile_out, void *stream);
/* Blocks (work units) the compositing launches of the calling thread's most recent gms_rasterize_forward were sized for. */
int64_t gms_last_launched_units(void);
/* Which compositing implementation the calling thread's most recent gms_rasterize_forward launched: 1 = the micro-tile kernels
 * (blend_micro.hip: n_contrib holds positions in a 4x4 block's pre-filtered list), 0 = the quadrant kernels (blend.hip: positions
 * in the tile's list).  The decision is the library's (GMS_MICRO, the frame's binning capacity -- the instance count itself on an
 * overflow re-run); bindings report it instead of re-deriving it. */
int32_t gms_last_used_micro(void);

typedef struct GmsRasterBackwardArgs {
    int32_t P, D, M, width, height;
    int64_t num_rendered;             /* value returned by the forward call */
    int64_t binning_capacity;         /* instance capacity the forward's binning buffer was laid out for:
                                         the capacity hint when one was given and was sufficient, else
                                         max(num_rendered, 1) */
    const float *background;
    const float *means3D, *shs, *shs_rest, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
    const float *viewmatrix, *projmatrix, *campos;
    float scale_modifier, tan_fovx, tan_fovy;
    int32_t antialiasing, debug;
    const int32_t *radii;             /* [P] from forward */
    const void *geom_buffer;          /* the three scratch buffers of the forward call */
    const void *binning_buffer;
    const void *image_buffer;
    const float *dL_dout_color;       /* [3,H,W] */
    const float *dL_dout_invdepth;    /* [1,H,W] or NULL */
    /* scratch (device): [P,16] floats, 64-byte aligned, MUST be all zero on entry.  One 64-byte record per
     * Gaussian: the ten per-pixel partial sums of a (wave, splat) pair -- five moments of q = dL/dG*G (q dx, q dy,
     * q dx^2, q dx dy, q dy^2), sum q, the colour weights r, g, b and the inverse-depth weight -- leave the blend
     * kernel as ONE atomic instruction whose lanes all hit the same cache line; preprocess_bwd maps the moments
     * to the gradients of (mean2D, conic, opacity).  See grad_accum_rezero below. */
    float *grad_accum;
    /* outputs (device), all fully overwritten (no zero-fill needed) */
    float *dL_dmeans2D;    /* [P,3] gradient w.r.t. NDC mean (x,y), z column = 0 */
    float *dL_dopacity;    /* [P] */
    float *dL_dcolors;     /* [P,3] colors_precomp path: gradient of the colours.  SH path: ignored unless sh_factor_mode == 1
                              (below); in FACTORISED mode the call writes the clamp-masked dL/dcolour of this view here and does
                              NOT write dL_dsh / dL_dsh_rest (they may be NULL): dL/dsh = Y(dir) (x) dL/dcolour is formed later by
                              gms_sh_grad_expand, for one view or for the gathered factors of many (multi-GPU: 3 floats per Gaussian
                              per view travel instead of 48) */
    float *dL_dmeans3D;    /* [P,3] */
    float *dL_dcov3D;      /* [P,6] written only when cov3D_precomp != NULL (else may be NULL) */
    float *dL_dsh;         /* [P,M,3] written only when shs != NULL ([P,1,3] DC block in split storage) */
    float *dL_dsh_rest;    /* [P,M-1,3] written only when shs_rest != NULL */
    float *dL_dscales;     /* [P,3] written only when scales != NULL */
    float *dL_drotations;  /* [P,4] written only when rotations != NULL */
    /* 1: the call leaves grad_accum all zero again (the consuming kernel clears each record it read), so a caller
     * that keeps the buffer per (device, stream, P) never pays a 64*P-byte memset per backward; 0: left dirty. */
    int32_t grad_accum_rezero;
    /* work units reported by the forward (num_units_out), or 0: the backward launch is then sized from binning_capacity */
    int64_t num_units;
    /* factorised mode only: 1 = dL_dcolors has P+1 rows and the call writes the view's camera centre into row P (the factor then
     * carries everything gms_sh_grad_expand needs about its view: one buffer to exchange, no separate copy) */
    int32_t factor_campos_row;
    /* SH path only.  0 (default): dense SH gradient -- dL_dsh (and dL_dsh_rest in split storage) are required and written,
     * dL_dcolors is not touched.  1: factorised mode -- dL_dcolors is required and written, dL_dsh / dL_dsh_rest are not.
     * (ABI 2 switched on `dL_dcolors != NULL`; an explicit flag cannot be set by accident through a reused scratch pointer.) */
    int32_t sh_factor_mode;
    /* ABI 8 -- the backward of a frame that was rendered STRAIGHT FROM A MESH (GmsRasterForwardArgs.mesh + mesh_out_*): with `mesh` set (the
     * GmsMeshArgs of the forward; means3D / scales / rotations / opacities = the four tensors the forward stored) the thread of a Gaussian,
     * once it holds dL/dxyz, dL/dscale, dL/drotation, dL/dopacity in registers, carries them on through the face -> Gaussian
     * parameterization itself: dL/d_alpha, dL/d_scale, dL/d_opacity are stored, its share of the face's corner gradients is added to
     * mesh_dL_dvertices (which must be ALL ZERO on entry: GmsMeshArgs.prezero of the forward) -- the arithmetic of
     * gms_mesh_to_gaussians_backward with fused_activations, without the 44 bytes per Gaussian in between and without its launch.
     * dL_dmeans3D / dL_dscales / dL_drotations / dL_dopacity are then NOT written (may be NULL).  Uniform splats per face, at most 4 (every
     * splat adds to its face's three corners); not in deterministic mode (which sums corner gradients in a fixed order: use
     * gms_mesh_to_gaussians_backward). */
    const struct GmsMeshArgs *mesh;
    float *mesh_dL_dvertices;         /* [V,3], zero on entry, accumulated with float atomics */
    float *mesh_dL_dalpha;            /* [P,3] */
    float *mesh_dL_dscale;            /* [P]   */
    float *mesh_dL_d_opacity;         /* [P]   */
} GmsRasterBackwardArgs;

int32_t gms_rasterize_backward(const GmsRasterBackwardArgs *args, void *stream);

/* SH gradient from per-view colour-gradient factors (see dL_dcolors above):
 *   dL_dsh[i][k][c] (+)= sum_v Y_k(normalize(means3D[i] - campos[v])) * factors[v][i][c],  views in index order.
 * With V = 1 and the factor of the same backward call this equals the dense dL_dsh of gms_rasterize_backward bit for bit.
 * The reference has no counterpart (it is single-GPU, train.py:90-92 pops one camera per step); the dense result it
 * replaces is the dL_dsh of submodules/diff-gaussian-rasterization (computeColorFromSH backward), SURVEY.md A.5. */
typedef struct GmsShGradExpandArgs {
    int32_t P, D, M;           /* Gaussians, active SH degree (0..3), coefficients per Gaussian in the destination rows */
    int32_t V;                 /* views */
    const float *means3D;      /* [P,3] */
    const float *campos;       /* [V,3] camera centres (device) */
    const float *factors;      /* [V,P,3] clamp-masked dL/dcolour of every view (device); view v starts at factors + v*factor_stride */
    int64_t factor_stride;     /* floats between consecutive views; 0 = 3*P (densely packed) */
    float *dL_dsh;             /* [P,M,3]; or the [P,1,3] DC block when dL_dsh_rest is given */
    float *dL_dsh_rest;        /* [P,M-1,3] or NULL */
    int32_t accumulate;        /* 0: overwrite the destination, 1: add to it */
    int32_t debug;
} GmsShGradExpandArgs;
int32_t gms_sh_grad_expand(const GmsShGradExpandArgs *args, void *stream);

/* present[i] = 1 iff Gaussian i passes the near-plane test (view-space z > 0.2). */
int32_t gms_mark_visible(int32_t P, const float *means3D, const float *viewmatrix, const float *projmatrix,
                         uint8_t *present, void *stream);

/* ---- mesh-face -> Gaussian parameterization ------------------------------------------- */
#define GMS_ALPHA_RELU 0      /* alpha = relu(_alpha)+1e-8, L1-normalised  (gaussian_mesh_model.py:166-167) */
#define GMS_ALPHA_SOFTMAX 1   /* alpha = softmax(_alpha)                   (gaussian_flame_model.py:195) */

typedef struct GmsMeshArgs {
    int32_t F;                    /* faces (all meshes concatenated) */
    int32_t V;                    /* vertices */
    int64_t P;                    /* Gaussians = sum of splats over faces */
    int32_t splats_per_face;      /* >0: uniform S (P == F*S); 0: use face_splat_offset / splat_face */
    int32_t alpha_mode;
    const float *vertices;        /* [V,3] */
    const int64_t *faces;         /* [F,3] vertex indices (int64, as the reference stores them) */
    const int32_t *face_splat_offset; /* [F+1] CSR offsets into the splat axis (non-uniform case) or NULL */
    const int32_t *splat_face;    /* [P] face of each splat (non-uniform case) or NULL */
    const float *_alpha;          /* [P,3]  (the reference's [F,S,3] flattened) */
    const float *_scale;          /* [P] */
    int32_t fused_activations;    /* 1: the property getters of scene/gaussian_model.py:95-101 are fused in:
                                     forward also writes exp(scaling) and normalize(rotation); backward takes
                                     the gradients w.r.t. THOSE instead of (scaling, rotation) */
    const float *_opacity;        /* [P] raw opacities or NULL: fuses get_opacity = sigmoid(_opacity)
                                     (scene/gaussian_model.py:113-115) and its backward into the same kernels */
    float *prezero;               /* forward only, optional: scratch of prezero_count floats that extra blocks of the
                                     forward launch clear (the [V,3] buffer the backward will accumulate into) */
    int64_t prezero_count;
    int32_t vertex_grad_prezeroed;/* backward only: dL_dvertices is already all zero (cleared through `prezero`), so the
                                     per-splat and per-face parts run as ONE launch */
} GmsMeshArgs;

/* Outputs: alpha [P,3] (normalised barycentrics, kept because save_ply / the animated renderer
 * read `pc.alpha`), xyz [P,3], scaling [P,3] = log(relu(_scale*s)+eps), rotation [P,4] quaternion. */
int32_t gms_mesh_to_gaussians_forward(const GmsMeshArgs *args, float *alpha, float *xyz, float *scaling,
                                      float *rotation, float *scaling_activated /* [P,3] or NULL */,
                                      float *rotation_unit /* [P,4] or NULL */,
                                      float *opacity_activated /* [P] or NULL; needs args->_opacity */, void *stream);

/* Gradients of (xyz, scaling, rotation[, sigmoid(_opacity)]) -> (vertices, _alpha, _scale[, _opacity]).  Every
 * output is fully overwritten: dL_dvertices [V,3] is cleared by the first kernel and accumulated with atomics by
 * the second; dL_dalpha [P,3], dL_dscale [P] and dL_d_opacity [P] (NULL = not requested) are plain stores. */
int32_t gms_mesh_to_gaussians_backward(const GmsMeshArgs *args, const float *dL_dxyz, const float *dL_dscaling,
                                       const float *dL_drotation, const float *dL_dopacity_activated /* or NULL */,
                                       float *dL_dvertices, float *dL_dalpha, float *dL_dscale,
                                       float *dL_d_opacity /* or NULL */, void *stream);

/* ---- exact 3-NN mean squared distance ---------------------------------
 * Replaces the un-vendored `simple_knn._C.distCUDA2(points)` (.gitmodules:1-3) called at
 * scene/gaussian_model.py:134 and games/flat_splatting/scene/flat_gaussian_model.py:47 to size the initial
 * Gaussians: out[i] = mean of the squared distances from point i to its 3 nearest OTHER points
 * (coincident points count with distance 0; with N < 4 the mean runs over the N-1 points there are).
 * `workspace` is caller-owned device scratch of at least gms_knn_workspace_bytes(N) bytes. */
size_t gms_knn_workspace_bytes(int32_t N);
int32_t gms_knn_mean_dist2(int32_t N, const float *points /* [N,3] */, float *out /* [N] */, void *workspace,
                           size_t workspace_bytes, void *stream);

/* ---- fused L1 + SSIM photometric loss ---------------------------------
 * Replaces train.py:106-107 built on utils/loss_utils.py:17-18 (l1_loss) and :33-63 (ssim: 11x11 Gaussian window,
 * sigma 1.5, zero padding, C1=0.01^2, C2=0.03^2, mean over every pixel of every plane):
 *     value = w_l1 * mean|img-gt| + w_ssim * mean(ssim_map(img, gt)) + bias
 * The training loss is (w_l1, w_ssim, bias) = (1-lambda, -lambda, lambda); ssim() alone is (0, 1, 0). */
typedef struct GmsLossArgs {
    int32_t planes;               /* channels x batch: the window is applied per plane (conv2d groups=channel) */
    int32_t height, width;
    const float *img;             /* [planes,H,W] the rendered image (differentiated) */
    const float *gt;              /* [planes,H,W] */
    float w_l1, w_ssim, bias;
} GmsLossArgs;

/* floats needed in `partials` (per-block sums, reduced in a fixed order: the value is deterministic) */
size_t gms_l1_ssim_partials(int32_t planes, int32_t height, int32_t width);
/* out[3] = {value, mean|img-gt|, mean ssim}.  dmaps [3,planes,H,W] receives the derivatives of the SSIM map
 * w.r.t. its window moments for the backward pass; NULL when no gradient is needed. */
int32_t gms_l1_ssim_forward(const GmsLossArgs *args, float *dmaps, float *partials, float *out, void *stream);
/* dL_dimg [planes,H,W] = dL_dvalue * dvalue/dimg; dL_dvalue is a DEVICE scalar (NULL = 1.0). */
int32_t gms_l1_ssim_backward(const GmsLossArgs *args, const float *dmaps, const float *dL_dvalue, float *dL_dimg,
                             void *stream);

/* ---- multi-tensor Adam step --------------------------------------------
 * Replaces `gaussians.optimizer.step()` (train.py:147) for torch.optim.Adam(groups, lr=0.0, eps=1e-15) as built
 * by training_setup() (games/mesh_splatting/scene/gaussian_mesh_model.py:174-183): every tensor of every group in
 * one launch (per GMS_ADAM_MAX_TENSORS tensors).  `step` is the 1-based step count AFTER the increment, `lr` the
 * group's learning rate.  No amsgrad / weight decay / maximize (the reference uses none). */
#define GMS_ADAM_MAX_TENSORS 16
typedef struct GmsAdamTensor {
    float *param;                 /* [n] updated in place */
    const float *grad;            /* [n] */
    float *exp_avg;               /* [n] updated in place */
    float *exp_avg_sq;            /* [n] updated in place */
    int64_t n;
    float lr;
    int32_t step;
} GmsAdamTensor;
int32_t gms_adam_step(const GmsAdamTensor *tensors /* HOST array */, int32_t count, double beta1, double beta2, double eps,
                      void *stream);

/* ---- per-kernel timing (HIP events on the launch stream; off by default) ------------------
 * When enabled every kernel launch made by this library is bracketed by two hipEvents on the
 * caller's stream.  gms_profile_read() synchronises the recorded events and returns the summed
 * duration and launch count per kernel since the last reset.  Used by bench.py for the live
 * roofline figure; costs two event records per launch, so keep it off in timed regions. */
#define GMS_K_PREPROCESS_FWD 0
#define GMS_K_TILE_SCAN 1
#define GMS_K_EMIT 2
#define GMS_K_TILE_SORT 3
#define GMS_K_BLEND_FWD 4
#define GMS_K_BLEND_BWD 5
#define GMS_K_PREPROCESS_BWD 6
#define GMS_K_MESH_FWD 7
#define GMS_K_MESH_BWD_SPLAT 8
#define GMS_K_MESH_BWD_FACE 9
#define GMS_K_BLEND_HEAD 10
#define GMS_K_BLEND_FINALIZE 11
#define GMS_K_LOSS_FWD 12
#define GMS_K_LOSS_BWD 13
#define GMS_K_ADAM 14
#define GMS_K_SH_EXPAND 15
#define GMS_K_MICRO_FILTER 16
#define GMS_K_COUNT 17
void gms_profile_enable(int32_t on);
void gms_profile_reset(void);
int32_t gms_profile_read(int32_t kernel_id, double *total_ms, int64_t *launches);
const char *gms_profile_kernel_name(int32_t kernel_id);
/* Microseconds a bracketing event pair adds to a launch it times (measured on `stream` with a kernel that times itself by the
 * device's wall clock, averaged over `reps` launches; < 0 on failure).  A launch's own ramp-up and drain are part of that figure,
 * so durations corrected by it are a lower bound: bench.py subtracts it from every per-kernel average (without it the small kernels
 * read up to 32 % long and the table sums to more than the step). */
double gms_profile_event_overhead_us(void *stream, int32_t reps);

/* ---- introspection --------------------------------------------------------------------- */
/* Host time gms_rasterize_forward spent waiting for the instance count N since the last reset (the one place the
 * host blocks on the device): near zero when the host is the bottleneck, about one step when the GPU is. */
void gms_wait_stats(double *total_ms, int64_t *calls, int32_t reset);
/* Instances in the deepest tile of the calling thread's most recent gms_rasterize_forward (drives the sort's pass count). */
int64_t gms_last_deepest_tile(void);
int32_t gms_abi_version(void);
/* Text of the last error on the calling thread ("" if none). */
const char *gms_last_error(void);
/* Byte sizes of the scratch buffers for given problem sizes (what the callbacks will be asked for). */
size_t gms_geom_bytes(int32_t P);
size_t gms_image_bytes(int32_t width, int32_t height);
/* Byte offset, inside the image scratch buffer, of n_contrib [H*W] uint32 (1-based list position of the last splat each
 * pixel composited): lets a caller sum the per-pixel walk lengths ("interactions") after a forward. */
size_t gms_image_n_contrib_offset(int32_t width, int32_t height);
/* Byte offset, inside the image scratch buffer, of the frame's counts: uint32[4] = {instances N, deepest tile, work units, segment length}. */
size_t gms_image_counts_offset(int32_t width, int32_t height);
size_t gms_binning_bytes(int64_t num_instances, int32_t width, int32_t height);

/* ---- deterministic-reduction mode (SURVEY.md section 5 "race detection", section 7 hard part 1) ---------------------------
 * The reference's CUDA rasterizer accumulates gradients with float atomics (run-to-run noise of
 * 1e-7..1e-6 relative) and so does this library by default (the blend kernels' per-Gaussian records, the vertex gradient of the
 * mesh op).  With the mode ON every floating-point sum of gms_rasterize_backward and gms_mesh_to_gaussians_backward runs in a
 * FIXED order and no float atomic is issued, so two runs on the same inputs give bit-identical gradients:
 *   - compositing backward: each wave keeps its own LDS table, the rows of a wave add in row order, a unit leaves ONE partial
 *     record per (Gaussian, tile) instance (quadrant kernels: one per instance and 8x8 quadrant) with plain stores;
 *   - a reduction kernel sums a Gaussian's partial records in the order of its tile rectangle (y outer, x inner), finding each
 *     instance in the tile's sorted list by binary search on its (depth, id) key;
 *   - mesh op: per-(face, corner) gradients are stored, and each vertex sums its incident corners in ascending corner index.
 * Callers should also pass binning_capacity_hint = 0 in this mode (both bindings do): with a hint, the segment length and the
 * choice between the micro-tile and the quadrant kernels follow the hint -- the history of earlier frames -- and a frame's sums
 * are then grouped differently from the same frame rendered first.
 * Scratch comes from library-owned buffers (64 B per instance; 256 B with the quadrant kernels; 36 B per face).  Slower (the
 * default path's cost is stated in DESIGN.md); meant for tests, debugging and bit-reproducible training runs.
 * Default: off, or the environment variable GAMES_HIP_DETERMINISTIC=1 read at first use; process-wide. */
void gms_set_deterministic(int32_t on);
int32_t gms_get_deterministic(void);

/* ---- fault injection (test infrastructure of the PARITY CRITERION, not of the kernels) -------------------------------
 * tests/test_gpu_negative_controls.py must show that the gradient criterion of tests/_util.py can FAIL: a deliberately
 * wrong backward has to trip it.  `fault` selects one defect for the calling process until reset to 0 (the default).  Only this
 * call switches a fault on: no environment variable does (a stray one must not be able to corrupt gradients).  The faulty code lives in separate
 * template instantiations of the kernels: the production instantiations contain no fault branch.
 *   1  blend_bwd: the second moment sum(q dx^2) of every 1000th Gaussian (id % 1000 == 0) is scaled by 1 + 2e-3
 *   2  blend_bwd: a unit that restarts the back-to-front recurrence at a segment boundary drops the colour composited
 *      behind it (the suffix of the later segments)
 *   3  the gradient records are NOT cleared after preprocess_bwd consumed them: the next backward of this (device,
 *      stream, P) adds the previous frame's moments to its own
 *   4  preprocess_bwd: dL/dscale of every 1000th Gaussian is scaled by 1 + 2e-3
 *   5  preprocess_bwd: dL/dscale (all three components) of every 100th Gaussian is scaled by 1 + 1.3e-3 -- a defect between the
 *      1e-3 tolerance and the 2e-3 of faults 1 / 4, on 1 % of the rows
 * These five are ALL the library contains; any other value means 0.  (Until round 4 a sixth value, 9, selected a wrong-results
 * timing experiment of the micro-tile backward; it no longer exists in any build.) */
void gms_set_fault(int32_t fault);
int32_t gms_get_fault(void);

/* ---- upstream-quirk switch (parity hygiene; DESIGN.md section 2 "unverifiable") --------------------
 * cov3D = R diag(mod s)^2 R^T, so the derivative with respect to the scale carries the factor mod = scale_modifier, and that is
 * what this library returns by default.  The public upstream CUDA backward (computeCov3D: `dL_dscale = dot(Rt[k], dL_dMt[k])` with
 * s = mod * scale) is believed to leave that factor out; the module is absent from the reference tree
 * (submodules/diff-gaussian-rasterization, .gitmodules:4-6), so it cannot be checked here.  With the switch on -- this call, or
 * GMS_UPSTREAM_SCALE_MOD_GRAD=1 in the environment at first use -- dL/dscale is returned WITHOUT the factor.  Inert in GaMeS:
 * every render() of the reference passes scale_modifier = 1.0 (renderer/gaussian_renderer/__init__.py:25). */
void gms_set_upstream_scale_mod_grad(int32_t on);
int32_t gms_get_upstream_scale_mod_grad(void);

#ifdef __cplusplus
}
#endif
#endif /* GMSPLAT_H */
