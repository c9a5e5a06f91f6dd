"""Helpers shared by the GPU parity tests: run the HIP path and the CPU oracle on the same
inputs and compare with the discontinuity-aware rules described in DESIGN.md (parity section)."""
import math

import numpy as np
import torch

from oracle import gs_oracle


def settings_kwargs(cam, bg, sh_degree=3, antialiasing=False, scale_modifier=1.0):
    return dict(image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                bg=bg, scale_modifier=scale_modifier, viewmatrix=cam.world_view_transform,
                projmatrix=cam.full_proj_transform, sh_degree=sh_degree, campos=cam.camera_center,
                prefiltered=False, debug=False, antialiasing=antialiasing)


def hip_render(inputs: dict, kw: dict, device="cuda", grad_color=None, grad_invdepth=None, need_grad=True):
    """inputs: means3D, opacities and (shs | colors_precomp), (scales, rotations | cov3D_precomp) as CPU tensors."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device(device)
    t = {k: (v.to(dev).float().detach().clone().requires_grad_(need_grad) if v is not None else None)
         for k, v in inputs.items()}
    kwd = dict(kw)
    for k in ("bg", "viewmatrix", "projmatrix", "campos"):
        kwd[k] = kwd[k].to(dev).float()
    rs = GaussianRasterizationSettings(**kwd)
    means2D = torch.zeros_like(t["means3D"], requires_grad=need_grad)
    color, radii, invd = GaussianRasterizer(rs)(
        means3D=t["means3D"], means2D=means2D, opacities=t["opacities"], shs=t.get("shs"),
        colors_precomp=t.get("colors_precomp"), scales=t.get("scales"), rotations=t.get("rotations"),
        cov3D_precomp=t.get("cov3D_precomp"))
    out = dict(color=color.detach().cpu().numpy(), radii=radii.cpu().numpy(), invdepth=invd.detach().cpu().numpy())
    if grad_color is not None:
        loss = (color * torch.as_tensor(grad_color, device=dev, dtype=torch.float32)).sum()
        if grad_invdepth is not None:
            loss = loss + (invd * torch.as_tensor(grad_invdepth, device=dev, dtype=torch.float32)).sum()
        loss.backward()
        g = {k: (v.grad.detach().cpu().numpy() if v is not None and v.grad is not None else None) for k, v in t.items()}
        g["means2D"] = means2D.grad.detach().cpu().numpy()
        out["grads"] = g
    torch.cuda.synchronize()
    return out


def oracle_render(inputs: dict, kw: dict, grad_color=None, grad_invdepth=None, precision="f32", amb_policy=0):
    okw = {k: v for k, v in kw.items() if k not in ("prefiltered", "debug")}
    o = gs_oracle.rasterize(**{k: v for k, v in inputs.items() if v is not None}, **okw, precision=precision, amb_policy=amb_policy)
    res = dict(color=o.color, radii=o.radii, invdepth=o.invdepth, N=o.N, interactions=o.interactions,
               details=o.state.details())
    if grad_color is not None:
        g = gs_oracle.backward(o, grad_color, grad_invdepth)
        res["grads"] = dict(means3D=g["means3D"], means2D=g["means2D"], shs=g["sh"], colors_precomp=g["colors_precomp"],
                            opacities=g["opacities"], scales=g["scales"], rotations=g["rotations"],
                            cov3D_precomp=g["cov3D_precomp"])
        res["sh_factor"] = g["sh_factor"]      # SH path: clamp-masked dL/dcolour per Gaussian (factorised SH gradient)
    return res


def forward_report(hip, ora, W, H, input_rounding=False):
    """Discontinuity-aware comparison.  Returns dict of statistics.  `input_rounding`: the two sides' rasterizer INPUTS
    differ by float32 rounding (each ran its own mesh->Gaussian stage): pixels whose skip decision is within that input
    rounding (oracle pix_ambig bit 1, conditioning-aware) are treated as ambiguous too; with identical inputs only the
    exp()-rounding band (bit 0) is."""
    d = ora["details"]
    mism = hip["radii"] != ora["radii"]
    unexplained = mism & ~((d["gauss_ambig"] & 1).astype(bool))       # bit 0: discrete per-Gaussian decision (radius, rect, near plane)
    amb = ((d["pix_ambig"] & (3 if input_rounding else 1)) != 0).copy()
    gx, gy = (W + 15) // 16, (H + 15) // 16
    # pixels of tiles touched by Gaussians whose discrete footprint differs are excluded too
    for i in np.nonzero(mism)[0]:
        r = max(int(hip["radii"][i]), int(ora["radii"][i]))
        x, y = d["xy"][i]
        x0, x1 = max(0, int((x - r) // 16) - 1), min(gx, int((x + r) // 16) + 2)
        y0, y1 = max(0, int((y - r) // 16) - 1), min(gy, int((y + r) // 16) + 2)
        amb[y0 * 16:y1 * 16, x0 * 16:x1 * 16] = True
    diff = np.abs(hip["color"] - ora["color"]).max(axis=0)
    ddiff = np.abs(hip["invdepth"][0] - ora["invdepth"][0])
    clean = ~amb
    mse = float(np.mean((hip["color"] - ora["color"]) ** 2))
    return dict(radii_mismatch=int(mism.sum()), radii_unexplained=int(unexplained.sum()), amb_frac=float(amb.mean()),
                max_clean=float(diff[clean].max()) if clean.any() else 0.0,
                max_amb=float(diff[amb].max()) if amb.any() else 0.0,
                max_invdepth_clean=float(ddiff[clean].max()) if clean.any() else 0.0,
                psnr=float(10 * math.log10(1.0 / mse)) if mse > 0 else float("inf"))


GRAD_REL = 1e-3          # north_star: 1e-3 relative on gradients
# An outlier is float32 conditioning if HIP is within K x the float32 ORACLE's own error against the float64 oracle on the
# same row.  "The float32 oracle's error" = the largest over its float32 REALISATIONS: the default build (float32 terms
# summed in double: per-term rounding only), the float32-accumulator build (libgs_oracle_f32acc: the sums themselves in
# float32, as any float-atomics implementation -- the reference's CUDA kernels, these HIP kernels -- has them) and the
# FMA-contracted backward (libgs_oracle_f32fma: the same formulas with a*b+c fused, which is what nvcc and hipcc both emit),
# and the default build on inputs nudged by one float32 ulp (`f32_realisations`).
# The affected rows are thin, long splats whose cov2D gradient is a difference of terms ~5000x its size.
# Round 3: every constant below was tightened to <= 2x the maximum observed over the round-2 report (27 comparisons incl.
# three config-5-size frames, profiles/r02_parity_report.jsonl) and the fuzz sweeps (400 + 1000 cases on the round-3
# kernels); DESIGN.md section 2 holds the table (constant, the failure that introduced it, observed maximum).
# tests/test_gpu_negative_controls.py shows the criterion FAILS for four injected defects (gmsplat.h, gms_set_fault).
ADJUDICATE_K = 8.0                # observed worst ratio 3.4 (scales, config-5 size), 3.9 (800-case fuzz) with the realisations below
UNEXPLAINED_PER_MILLION = 1.0     # observed 0
RARE_FRAC = 5e-4                  # explained outliers per tensor: observed <= 5.2e-5 (scales); small tensors: <= RARE_MIN entries
RARE_MIN = 8
ROW_FRAC = 1e-2                   # entries taking the excused / alternate-outcome rules: suite <= 1.3e-5; fuzz seed 7197 (faint splats, alpha near 1/255 everywhere): 26 of 4500
Q_MIN_SIZE = 500                  # the 0.999 quantile is asserted for every tensor with at least this many entries
# Conditioning-relative caps.  On 0.7 % of random scenes (faint, large, elongated splats) the float32 ORACLE ITSELF is not
# within 1e-3 of the float64 one: e.g. fuzz seed 5179, rotation gradients: 27 of 6000 entries of the float32 realisations
# beyond 1e-3 from float64, 0.999-quantile 2.8e-3; HIP against the float32 oracle: 15 entries, 2.4e-3.  There the fixed
# caps above cannot hold for ANY float32 implementation, so the cap becomes relative to what the oracle's own float32
# realisations do on the same tensor (`ref_outliers`, `ref_q`: measured without the HIP output):
COND_COUNT = 2.0                  # outliers <= COND_COUNT x ref_outliers   (observed <= 1.25 x where more than RARE_MIN entries are involved)
COND_Q = 3.0                      # HIP's 0.999-quantile error AGAINST FLOAT64 <= COND_Q x ref_q   (observed <= 1.89 x, 600-case fuzz)


def nudged_inputs(inputs, seed):
    """The same scene with every coordinate of means3D / scales / rotations moved to a NEIGHBOURING float32 (random
    direction, seeded): to float64 the gradient changes by ~1e-5 of its scale; to a float32 evaluation it is a fresh
    rounding realisation of every intermediate."""
    rng = np.random.default_rng(seed)
    out = dict(inputs)
    for k in ("means3D", "scales", "rotations"):
        if inputs.get(k) is not None:
            a = np.asarray(inputs[k].detach().cpu().numpy() if torch.is_tensor(inputs[k]) else inputs[k], np.float32)
            out[k] = torch.from_numpy(np.nextafter(a, np.where(rng.integers(0, 2, a.shape) > 0, np.inf, -np.inf).astype(np.float32)))
    return out


def f32_realisations(inputs, kw, grad_color, grad_invdepth=None, nudged=(1, 2)):
    """Gradients of further float32 realisations of the oracle (same algorithm, different rounding), the noise scale of
    `assert_grads(go32acc_fn=...)`: float32 accumulators; the FMA-contracted backward where the host CPU has FMA3; and the
    default build on inputs nudged by one float32 ulp (seeds `nudged`).  The last is the one that does not depend on luck:
    on fuzz seed 5337 (one elongated Gaussian) the default and accumulator builds are within 1e-4 of float64 on the
    rotation gradient, every nudged realisation and the contracted one are 1.8e-3 ... 3.5e-3 away."""
    out = [oracle_render(inputs, kw, grad_color, grad_invdepth, precision="f32acc")["grads"]]
    if gs_oracle.has_fma():
        out.append(oracle_render(inputs, kw, grad_color, grad_invdepth, precision="f32fma")["grads"])
    for sd in nudged:
        out.append(oracle_render(nudged_inputs(inputs, sd), kw, grad_color, grad_invdepth)["grads"])
    return out


def fuzz_case(seed):
    """The random scene / camera / settings of tools/fuzz_parity.py for one seed -> (inputs, kw, tag, rng): shared by the
    sweep, tools/diag_case.py and the CPU self-checks of the criterion (tests/test_oracle_raster.py)."""
    from games_hip import synthetic as syn
    rng = np.random.default_rng(seed)
    P = int(rng.choice([1, 7, 100, 1500, 6000, 20000]))
    W, H = int(rng.integers(17, 300)), int(rng.integers(17, 220))
    deg = int(rng.integers(0, 4))
    aa = bool(rng.integers(0, 2))
    lo = float(rng.choice([0.002, 0.01, 0.05])); hi = lo * float(rng.choice([2, 10, 40]))
    op_lo = float(rng.choice([0.01, 0.1, 0.6])); op_hi = min(0.999, op_lo + float(rng.choice([0.05, 0.4])))
    sc = syn.random_scene(P, seed=seed, scale_lo=lo, scale_hi=hi, opacity_lo=op_lo, opacity_hi=op_hi)
    cam = syn.orbit_camera(int(rng.integers(0, 8)), width=W, height=H, radius=float(rng.choice([1.5, 3.0, 6.0])))
    bg = torch.tensor(rng.uniform(0, 1, 3), dtype=torch.float32)
    kw = settings_kwargs(cam, bg, antialiasing=aa, sh_degree=deg, scale_modifier=float(rng.choice([1.0, 0.6, 1.8])))
    inputs = dict(means3D=sc.means3D, opacities=sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    tag = f"P={P} {W}x{H} deg={deg} aa={aa} scale=[{lo},{hi}] op=[{op_lo},{op_hi}]"
    return inputs, kw, tag, rng


def excused_rows(details):
    """Rows (Gaussians) that own a (pixel, Gaussian) skip / stop decision inside exp() rounding (oracle gauss_ambig bit 1):
    an independent float32 implementation may include or drop that one pixel's term of their gradient."""
    return (np.asarray(details["gauss_ambig"]) & 2) != 0


def alt_oracles(inputs, kw, grad_color, grad_invdepth, details):
    """(rows, fn) for `assert_grads(alt=...)`.  rows: Gaussians composited in a pixel that holds a within-rounding decision
    of ANY Gaussian (gauss_ambig bit 2) -- the outcome changes that pixel's term for all of them (the transmittance behind
    the pair, the colour behind for those in front).  fn() -> the oracle's gradients with every such decision forced IN
    and forced OUT: a row of `rows` may agree with either instead of with the as-computed oracle."""
    rows = (np.asarray(details["gauss_ambig"]) & 4) != 0
    return rows, (lambda: [oracle_render(inputs, kw, grad_color, grad_invdepth, amb_policy=p)["grads"] for p in (1, -1)])


def grad_report(gh, go, q=0.999, go64=None, excuse=None, go32acc=None, alt_rows=None, alts=None, K=None):
    """Per-tensor error statistics relative to the tensor's own scale, plus the HARD criteria:

      * `zero_violation`: the oracle's gradient tensor is identically zero but the HIP one is not;
      * `outliers`: entries with |hip - o32| > 1e-3 (|o32| + 1e-3 max|o32|);
      * `unexplained`: outliers that are NOT float32 conditioning.  With the float64 oracle `go64` (same algorithm in
        double: oracle/gs_oracle.c built with ORACLE_DOUBLE) an outlier is explained when the HIP value is as close to
        the float64 truth as the float32 ORACLE itself is, up to a factor K (different summation order, float atomics):
            |hip - o64| <= K max_row|o32 - o64| + 1e-3 (|o64| + 1e-3 max|o64|)
        (the oracle's error is taken as the largest over the components of the same row).
        Without `go64` every outlier counts as unexplained.
    Tests assert unexplained == 0 and zero_violation == False: a bound on EVERY entry, not a quantile."""
    rep = {}
    for k, b in go.items():
        a = gh.get(k)
        if a is None or b is None or b.size == 0:
            continue
        a = np.asarray(a).reshape(b.shape)
        scale = float(np.abs(b).max())
        if scale == 0:
            mx = float(np.abs(a).max())
            rep[k] = dict(scale=0.0, max_abs=mx, q_rel=0.0, max_rel=0.0, frac_bad=0.0, outliers=0, unexplained=0,
                          zero_violation=bool(mx != 0.0), worst_ratio=0.0, excused=0, size=int(b.size))
            continue
        err = np.abs(a - b)
        rel = err / (np.abs(b) + 1e-3 * scale)      # 1e-3 relative with an absolute floor of 1e-3*max|g|
        bad = rel > GRAD_REL
        n_exc = 0
        if excuse is not None and bad.any() and b.shape[0] == excuse.shape[0]:
            # rows (Gaussians) with a pixel-level skip decision inside exp() rounding (oracle gauss_ambig bit 1): an
            # implementation may include or drop that pixel's term; bounded at 5 % of the tensor's largest gradient instead of at 1e-3
            row_exc = np.broadcast_to(excuse.reshape((-1,) + (1,) * (b.ndim - 1)), b.shape)
            n_exc = int((bad & row_exc).sum())
            # one fringe pixel (alpha = 1/255) in or out: large RELATIVE to a small, self-cancelling gradient (the mean
            # gradient of a symmetric splat sums to ~0), never comparable to the largest gradient of the tensor
            assert float(err[bad & row_exc].max() if n_exc else 0.0) <= 0.05 * scale, (k, "excused outlier too large")
            bad = bad & ~row_exc
        n_alt = 0
        if alt_rows is not None and alts and bad.any() and b.shape[0] == alt_rows.shape[0]:
            # rows sharing a pixel with a within-rounding decision: accept an entry that agrees (same 1e-3 rule) with the
            # oracle under one of the two forced outcomes, or lies between them (several such pixels on one row)
            row_alt = np.broadcast_to(alt_rows.reshape((-1,) + (1,) * (b.ndim - 1)), b.shape)
            cand = [np.asarray(x[k], b.dtype).reshape(b.shape) for x in alts if x.get(k) is not None]
            if cand:
                tol = GRAD_REL * (np.abs(b) + 1e-3 * scale)
                lo_, hi_ = np.minimum.reduce(cand + [b]), np.maximum.reduce(cand + [b])
                agrees = (a >= lo_ - tol) & (a <= hi_ + tol)
                n_alt = int((bad & row_alt & agrees).sum())
                bad = bad & ~(row_alt & agrees)
        n_out = int(bad.sum())
        unexplained, worst = n_out, 0.0
        ref_n, ref_q, q64 = 0, 0.0, None
        if n_out and go64 is not None and go64.get(k) is not None:
            t = np.asarray(go64[k], np.float64).reshape(b.shape)
            s64 = float(np.abs(t).max())
            e_hip = np.abs(a.astype(np.float64) - t)[bad]
            # the float32 oracle's own error, taken per ROW (per Gaussian / vertex: conditioning -- a flat covariance, a
            # cancelling conic gradient -- is a property of the row, and the rounding of one component can be lucky)
            e32 = np.abs(b.astype(np.float64) - t)
            tol64 = np.abs(t) + 1e-3 * s64
            ref_n, ref_q = int((e32 / tol64 > GRAD_REL).sum()), float(np.quantile(e32 / tol64, q))
            # ... or of the other float32 realisations (float32 accumulators; FMA-contracted backward)
            for r in ([go32acc] if isinstance(go32acc, dict) else (go32acc or [])):
                if r.get(k) is not None:
                    e_r = np.abs(np.asarray(r[k], np.float64).reshape(b.shape) - t)
                    ref_n, ref_q = max(ref_n, int((e_r / tol64 > GRAD_REL).sum())), max(ref_q, float(np.quantile(e_r / tol64, q)))
                    e32 = np.maximum(e32, e_r)
            q64 = float(np.quantile(np.abs(a.astype(np.float64) - t) / tol64, q))
            if e32.ndim > 1:
                e32 = np.broadcast_to(e32.reshape(e32.shape[0], -1).max(axis=1).reshape((-1,) + (1,) * (e32.ndim - 1)), e32.shape)
            e_o32 = e32[bad]
            slack = GRAD_REL * (np.abs(t)[bad] + 1e-3 * s64)
            ok = e_hip <= (ADJUDICATE_K if K is None else K) * e_o32 + slack
            unexplained = int((~ok).sum())
            with np.errstate(divide="ignore", invalid="ignore"):
                ratio = np.where(e_o32 > 0, (e_hip - slack) / e_o32, np.where(e_hip > slack, np.inf, 0.0))
            worst = float(np.max(ratio)) if ratio.size else 0.0
        # (the quantile is also taken over the entries that did NOT need an excuse: on a tensor of a few hundred entries
        # one excused fringe-pixel row would otherwise BE the 0.999 quantile)
        clean_mask = np.ones(rel.shape, bool)
        if n_exc:
            clean_mask &= ~(row_exc & (rel > GRAD_REL))
        if n_alt:
            clean_mask &= ~(row_alt & agrees & (rel > GRAD_REL))
        q_clean = float(np.quantile(rel[clean_mask], q)) if clean_mask.any() else 0.0
        rep[k] = dict(scale=scale, max_abs=float(err.max()), q_rel=float(np.quantile(rel, q)), q_rel_clean=q_clean, max_rel=float(rel.max()),
                      frac_bad=float(bad.mean()), outliers=n_out, unexplained=unexplained, zero_violation=False,
                      worst_ratio=worst, excused=n_exc, alt_explained=n_alt, size=int(b.size),
                      ref_outliers=ref_n, ref_q=ref_q, q_rel64=q64)
    return rep


# Deterministic-reduction mode (include/gmsplat.h, GAMES_HIP_DETERMINISTIC=1): no float atomics, fixed summation order.  The two
# allowances that exist for the ORDER of float atomics are tightened there: K is halved to ADJUDICATE_K_STRICT and not a single
# entry may stay unexplained.  (K cannot go to 1: the fixed order is one more float32 realisation of the sums, and how far one
# realisation sits from float64 relative to the few others that were sampled has a tail of its own -- worst ratio 3.9 over 800
# default-mode fuzz scenes in round 3, 3.35 over 120 deterministic-mode scenes in round 4, both on scenes of ONE elongated Gaussian;
# K = 2 passed the suite and 119 of those 120 scenes.)
ADJUDICATE_K_STRICT = 4.0


def grad_fails(v, strict=False):
    """The assertions of `assert_grads` on one tensor's report, as a list of the rules it breaks (empty = passes)."""
    out = []
    if v["zero_violation"]:
        out.append("zero_violation")
    # explained outliers must stay rare: a fixed cap, or -- where the float32 oracle's own realisations leave the 1e-3 band
    # against float64 on this tensor -- COND_COUNT x their count
    if v["outliers"] > max(RARE_MIN, int(RARE_FRAC * v["size"]), int(COND_COUNT * v.get("ref_outliers", 0))):
        out.append("outliers")
    # entries that needed the excused-row or the alternate-outcome rule are counted and capped too
    if v.get("excused", 0) + v.get("alt_explained", 0) > max(RARE_MIN, int(ROW_FRAC * v["size"])):
        out.append("excused_rows")
    # the 0.999 quantile against the float32 oracle (below 8000 entries over the non-excused entries: for < 1000 entries
    # it is their maximum); where it is exceeded, HIP's quantile error against FLOAT64 must be within COND_Q x the float32
    # oracle's own
    q_ok = v["q_rel"] <= GRAD_REL or v["size"] < Q_MIN_SIZE or (v["size"] < 8000 and v.get("q_rel_clean", v["q_rel"]) <= GRAD_REL)
    if not q_ok and not (v.get("q_rel64") is not None and v["q_rel64"] <= COND_Q * v.get("ref_q", 0.0)):
        out.append("quantile")
    if v["unexplained"] > (0 if strict else int(UNEXPLAINED_PER_MILLION * 1e-6 * v["size"])):
        out.append("unexplained")
    return out


def assert_grads(gh, go, go64_fn=None, q=0.999, where="", excuse=None, go32acc_fn=None, alt=None, strict=False):
    """The gradient criterion of every parity test: quantile <= 1e-3 AND no unexplained outlier AND no non-zero gradient
    where the oracle's is identically zero AND explained / excused entries rare.  `go64_fn()` (lazy: only evaluated if
    some entry is an outlier) returns the float64 oracle's gradients.  Constants: top of this file; their history and
    observed maxima: DESIGN.md section 2; proof that the criterion can fail: tests/test_gpu_negative_controls.py.
    `strict` (deterministic-reduction mode): K = ADJUDICATE_K_STRICT, zero unexplained entries."""
    K = ADJUDICATE_K_STRICT if strict else None
    rep = grad_report(gh, go, q=q, excuse=excuse, K=K)
    if go64_fn is not None and any(v["outliers"] for v in rep.values()):
        alt_rows, alts = (alt[0], alt[1]()) if alt is not None and alt[0].any() else (None, None)
        rep = grad_report(gh, go, q=q, go64=go64_fn(), excuse=excuse, go32acc=go32acc_fn() if go32acc_fn is not None else None,
                          alt_rows=alt_rows, alts=alts, K=K)
    for k, v in rep.items():
        broken = grad_fails(v, strict=strict)
        assert not broken, (where, k, broken, v)
    _log_parity(where, rep)
    return rep


# ---- small entries (round-5 review, item 9).  The per-entry rule above is allclose(rtol = 1e-3, atol = 1e-6 max|g|) -- SURVEY.md A.6 -- so
# an entry between 1e-6 and 1e-3 of its tensor's largest is only held to 1e-6 max|g| ABSOLUTE: a defect confined to faint / far
# Gaussians (5 % on rows at 1e-5 of the maximum) is invisible to it.  This second check is purely RELATIVE on exactly that band, against
# the float64 oracle:   |hip - o64| <= K x noise(row) + 1e-3 |o64|,   noise(row) = the largest |o32* - o64| over the float32
# realisations of the oracle and the components of the row (what float32 conditioning does to that Gaussian; K as in rule (b)).  Rows that
# own or share a within-rounding skip / stop decision are left to rules (a), (a').  Observed on the oracle's own float32 build against
# the other realisations (suite scenes + fuzz seeds, tests/test_oracle_raster.py): 0 violations, worst ratio 1.1.
SMALL_LO, SMALL_HI = 1e-6, 1e-3
SMALL_PER_10K = 1.0               # default (float-atomics) mode: violations per 10 000 band entries; deterministic mode: none


def small_entry_report(gh, go64, reals, K=None, skip_rows=None):
    K = ADJUDICATE_K if K is None else K
    rep = {}
    for k, t in go64.items():
        a = gh.get(k)
        if a is None or t is None or np.size(t) == 0:
            continue
        t = np.asarray(t, np.float64)
        a = np.asarray(a, np.float64).reshape(t.shape)
        S = float(np.abs(t).max())
        if S == 0.0:
            continue
        band = (np.abs(t) >= SMALL_LO * S) & (np.abs(t) < SMALL_HI * S)
        if skip_rows is not None and t.shape[0] == skip_rows.shape[0]:
            band &= ~np.broadcast_to(skip_rows.reshape((-1,) + (1,) * (t.ndim - 1)), t.shape)
        if not band.any():
            continue
        noise = np.zeros_like(t)
        for r in reals:
            if r.get(k) is not None:
                noise = np.maximum(noise, np.abs(np.asarray(r[k], np.float64).reshape(t.shape) - t))
        if noise.ndim > 1:
            noise = np.broadcast_to(noise.reshape(noise.shape[0], -1).max(axis=1).reshape((-1,) + (1,) * (noise.ndim - 1)), noise.shape)
        err = np.abs(a - t)
        slack = GRAD_REL * np.abs(t)
        viol = band & (err > K * noise + slack)
        with np.errstate(divide="ignore", invalid="ignore"):
            ratio = np.where(noise > 0, (err - slack) / noise, np.where(err > slack, np.inf, 0.0))[band]
        rep[k] = dict(band=int(band.sum()), violations=int(viol.sum()), worst_ratio=float(ratio.max()),
                      median_rel=float(np.median(err[band] / np.abs(t[band]))), size=int(t.size))
    return rep


def assert_small_entries(gh, go64, reals, where="", strict=False, skip_rows=None):
    rep = small_entry_report(gh, go64, reals, K=ADJUDICATE_K_STRICT if strict else None, skip_rows=skip_rows)
    for k, v in rep.items():
        allowed = 0 if strict else int(SMALL_PER_10K * 1e-4 * v["band"])
        assert v["violations"] <= allowed, (where, k, "small entries", v)
    return rep


def faint_row_defect(grads, go64, key, frac=0.01, factor=1.05):
    """The negative control of the small-entry check: the faintest `frac` of the rows of tensor `key` that are not numerically dead
    (largest entry >= 1e-6 of the tensor's largest) multiplied by `factor`.  Returns a copy of `grads`."""
    out = dict(grads)
    t = np.array(out[key], copy=True)
    S = float(np.abs(np.asarray(go64[key])).max())
    mag = np.abs(np.asarray(go64[key])).reshape(t.shape[0], -1).max(axis=1)
    rows = np.nonzero(mag >= SMALL_LO * S)[0]
    faint = rows[np.argsort(mag[rows])[:max(1, int(frac * len(rows)))]]
    t[faint] = t[faint] * factor
    out[key] = t
    return out


def _memo(fn):
    """Evaluate a lazy oracle (float64 run, float32 realisations, alternate outcomes) at most once across the two legs below."""
    if fn is None:
        return None
    box = []

    def get():
        if not box:
            box.append(fn())
        return box[0]
    return get


def assert_grads_both_modes(run_hip, go, go64_fn=None, q=0.999, where="", excuse=None, go32acc_fn=None, alt=None, small=False):
    """The suite's gradient gate since round 5.  PRIMARY: the kernels in deterministic-reduction mode
    (no float atomic, fixed summation order) under the STRICT criterion -- K = ADJUDICATE_K_STRICT, not one unexplained entry.
    SECONDARY: the default mode (float atomics, as the reference's CUDA kernels) under the default criterion, whose two allowances
    (K = 8, one unexplained entry per million) exist for the atomics' summation order alone.  `run_hip()` renders + differentiates
    with the current mode and returns a dict with "grads" (it is called once per leg).  Returns (default-mode output, its report)."""
    import diff_gaussian_rasterization as dgr
    go64_fn, go32acc_fn = _memo(go64_fn), _memo(go32acc_fn)
    if alt is not None:
        alt = (alt[0], _memo(alt[1]))
    was = dgr.deterministic()
    dgr.set_deterministic(True)
    try:
        h_det = run_hip()
    finally:
        dgr.set_deterministic(was)
    assert_grads(h_det["grads"], go, go64_fn, q=q, where=where + " [deterministic, strict]", excuse=excuse, go32acc_fn=go32acc_fn, alt=alt,
                 strict=True)
    h = run_hip()
    rep = assert_grads(h["grads"], go, go64_fn, q=q, where=where + " [atomics]", excuse=excuse, go32acc_fn=go32acc_fn, alt=alt)
    if small and go64_fn is not None and go32acc_fn is not None:
        # the purely relative check of the entries between 1e-6 and 1e-3 of each tensor's largest (both legs; needs the float64 run and
        # the float32 realisations whether or not the first criterion asked for them)
        skip = excuse if excuse is not None else None
        if alt is not None:
            skip = alt[0] if skip is None else (skip | alt[0])
        reals = list(go32acc_fn()) + [go]
        srep = assert_small_entries(h_det["grads"], go64_fn(), reals, where + " [deterministic, strict]", strict=True, skip_rows=skip)
        assert_small_entries(h["grads"], go64_fn(), reals, where + " [atomics]", skip_rows=skip)
        for k, v in srep.items():
            rep.setdefault(k, {})["small_band"] = v["band"]; rep[k]["small_worst_ratio"] = v["worst_ratio"]
    return h, rep


def _log_parity(where, rep):
    """Append the per-tensor maxima to gpurun_out/parity_report.jsonl (merged back from the GPU box): the numbers
    DESIGN.md quotes for max_rel come from here."""
    import json
    import os
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_report.jsonl"), "a") as f:
            f.write(json.dumps({"where": where, "tensors": {k: {m: v[m] for m in ("scale", "q_rel", "max_rel", "outliers",
                                                                                 "unexplained", "worst_ratio", "excused", "alt_explained", "size", "q_rel_clean") if m in v}
                                                             for k, v in rep.items()}}) + "\n")
    except OSError:
        pass
