"""CPU: the BOUNDS behind the micro-tile backward's 64-bit fixed-point gradient table (csrc/blend_micro.hip, "fixed-point gradient
table"; csrc/gms_blend.h::bwd_step, fx_exp, fx_field_base), restated in numpy float32 and attacked with adversarial 4x4 blocks.

The kernel adds a block's partial sum y of field f as round(y * 2^(47 - E_f)), E_f fixed per unit from
    |q| <= OP ((Cmax + |bg|) D1 + 5 Dd)        |dx| <= X, |dy| <= Y        colour weights <= 1
(OP the unit's largest opacity, Cmax its largest |colour component|, D1 its largest sum_c |dL/dpixel_c|).  The table cannot
overflow if |y| <= 2^E_f for every (block, splat, field); the conversion stays exact up to 2^51, i.e. 16x above that.  This test
runs the recurrence of bwd_step over random splat lists (opacities at the 0.99 clamp, colours of either sign, pixel gradients over
twelve decades, far-away centres) and checks the bound with NO allowance, and that the bound is not vacuous (it is reached within
2^-12 somewhere: a value that small below its bound still carries twelve bits above the rounding unit of 2^(E-48) ... 2^(E-47))."""
import numpy as np

F32 = np.float32


def fx_exp(x):
    """gms_blend.h::fx_exp: x < 2^fx_exp(x) for finite x >= 0 (biased exponent - 126)."""
    return int((np.float32(x).view(np.uint32) >> 23) & 0xff) - 126


def block_partial_sums(rng, n, signed_colours):
    """One 4x4 block, one list of n splats: the float32 recurrence of bwd_step, back to front; returns per-splat sums over the 16
    pixels of the nine fields and the unit-level maxima the kernel derives its exponents from."""
    op = np.where(rng.random(n) < 0.3, 1.0, rng.random(n) ** 3 + 1e-3).astype(F32).clip(0, 1)
    cen = (rng.normal(0, 1, (n, 2)) * 10.0 ** rng.uniform(-1, 2.5, (n, 1))).astype(F32)         # centres up to hundreds of pixels away
    pix = np.stack(np.meshgrid(np.arange(4), np.arange(4)), -1).reshape(16, 2).astype(F32) + F32(0.5)
    d = cen[:, None, :] - pix[None, :, :]                                                      # [n,16,2]
    sig = (10.0 ** rng.uniform(-0.5, 2.5, n)).astype(F32)
    G = np.exp(-(d ** 2).sum(-1) / (2 * sig[:, None] ** 2)).astype(F32)                        # [n,16] <= 1
    cmag = F32(10.0 ** rng.uniform(-2, 2))
    col = (rng.uniform(-1 if signed_colours else 0, 1, (n, 3)) * cmag).astype(F32)
    col[rng.integers(0, n)] = cmag * (1 if not signed_colours else rng.choice([-1, 1], 3))     # the maximum is attained
    bg = (rng.uniform(-1 if signed_colours else 0, 1, 3) * cmag * rng.choice([0.0, 1.0, 3.0])).astype(F32)
    dp = (rng.normal(0, 1, (16, 3)) * 10.0 ** rng.uniform(-9, 3)).astype(F32)                  # dL/dpixel over twelve decades
    alpha = np.minimum(F32(0.99), op[:, None] * G).astype(F32)
    act = alpha >= F32(1.0 / 255.0)
    alpha = np.where(act, alpha, F32(0))
    Tfinal = np.prod((1 - alpha).astype(F32), axis=0, dtype=F32)
    # backward, back to front (bwd_step)
    T = Tfinal.copy()
    acc = np.zeros((16, 3), F32)
    bgdot = (Tfinal * (dp @ bg)).astype(F32)
    sums = np.zeros((n, 9), F32)
    for e in range(n - 1, -1, -1):
        a = alpha[e]
        om = (F32(1) - a).astype(F32)
        rcp = (F32(1) / om).astype(F32)
        T = (T * rcp).astype(F32)
        w = (a * T).astype(F32)
        dLda = (((col[e][None, :] - acc) * dp).sum(-1, dtype=F32) * T - bgdot * rcp).astype(F32)
        acc = (a[:, None] * col[e][None, :] + om[:, None] * acc).astype(F32)
        q = np.where(act[e], G[e] * op[e], F32(0)).astype(F32) * dLda
        dx, dy = d[e, :, 0], d[e, :, 1]
        v = np.stack([q * dx, q * dy, q * dx * dx, q * dx * dy, q * dy * dy, q, w * dp[:, 0], w * dp[:, 1], w * dp[:, 2]], -1).astype(F32)
        sums[e] = v.sum(0, dtype=F32)
    maxima = dict(cmax=np.abs(col).max(), bgm=np.abs(bg).max(), D1=np.abs(dp).sum(-1).max(), X=np.abs(d[..., 0]).max(),
                  Y=np.abs(d[..., 1]).max(), OP=op.max())
    return sums, maxima


def field_exponents(m):
    eK = fx_exp(F32(32) * ((F32(m["cmax"]) + F32(m["bgm"])) * F32(m["D1"]))) + fx_exp(m["OP"])
    eX, eY, eCol = fx_exp(m["X"]), fx_exp(m["Y"]), fx_exp(F32(32) * F32(m["D1"]))
    #        mx       my       ca (dx^2)     cb (dx dy)      cc (dy^2)     k    r     g     b
    return [eK + eX, eK + eY, eK + 2 * eX, eK + eX + eY, eK + 2 * eY, eK, eCol, eCol, eCol]


def test_partial_sums_stay_under_their_unit_level_bounds():
    rng = np.random.default_rng(2025)
    worst = -1e9
    closest = -1e9
    for case in range(400):
        n = int(rng.integers(1, 48))
        sums, m = block_partial_sums(rng, n, signed_colours=bool(case % 2))
        E = field_exponents(m)
        assert np.isfinite(sums).all()
        for f in range(9):
            top = float(np.abs(sums[:, f]).max())
            if top > 0:
                slack = np.log2(top) - E[f]                 # must be <= 0: |y| <= 2^E
                worst = max(worst, slack)
                closest = max(closest, slack)
    assert worst <= 0.0, worst                               # no allowance: the bound holds as stated
    assert closest > -12.0, closest                          # ... and is within reach of real values: not a vacuous bound


def test_exponent_helper_is_a_strict_upper_bound():
    rng = np.random.default_rng(7)
    x = (rng.uniform(1, 2, 2000) * 2.0 ** rng.integers(-100, 100, 2000)).astype(F32)
    for v in x:
        e = fx_exp(v)
        assert float(v) < 2.0 ** e <= 2.0 * float(v) * (1 + 1e-7)
    assert fx_exp(0.0) == -126 and fx_exp(1.0) == 1 and fx_exp(0.99) == 0


def test_the_bound_is_nearly_attained_by_an_opaque_splat_over_aligned_pixel_gradients():
    """One splat at the 0.99 clamp covering the block with G ~ 1, its colour +Cmax in every channel, the background -Cmax, every
    pixel's gradient +D in every channel: |q| per pixel = op G (Cmax + |bg|) * 3 D = op G (Cmax + |bg|) D1, sixteen pixels add up,
    and the bound's 32 = 16 pixels x 2 leaves exactly the factor 2 (plus what fx_exp rounds up)."""
    cm, D = F32(0.75), F32(0.3)
    op, G = F32(1.0), F32(0.999)
    alpha = min(F32(0.99), op * G)
    Tfinal = F32(1) - alpha
    T = Tfinal / (F32(1) - alpha)
    dLda = (cm * 3 * D) * T - (Tfinal * (-cm * 3 * D)) / (F32(1) - alpha)
    y = 16 * float(G * op * dLda)
    m = dict(cmax=cm, bgm=cm, D1=3 * D, X=1.0, Y=1.0, OP=op)
    eK = field_exponents(m)[5]
    assert -3.0 < np.log2(y) - eK <= 0.0
