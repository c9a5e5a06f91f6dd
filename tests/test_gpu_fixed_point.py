"""GPU: the conversions of the micro-tile backward's 64-bit fixed-point gradient table (gms_blend.h: fx_from_float, fx_to_float;
blend_micro.hip describes the scheme), run BY THE DEVICE FUNCTIONS THE KERNEL INLINES through libgmsplat_testhooks.so.

fx_from_float(y, k) must be round-half-even(y * 2^k) exactly -- the float multiply by a power of two is exact, the double add of
1.5 * 2^52 rounds once -- for every |y * 2^k| <= 2^47 (what the kernel's bounds guarantee) and must stay exact up to 2^50;
fx_to_float must invert it to the nearest float.  Also checked: sums of up to 16 converted values equal the integer sum of the
exact roundings in any order (the table's order independence), and a zero converts to zero (the kernel adds without testing)."""
import ctypes
import os
from fractions import Fraction

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hook():
    lib = ctypes.CDLL(os.path.join(ROOT, "gaussian-mesh-splatting_amd", "lib", "libgmsplat_testhooks.so"))
    lib.gms_test_fixed_point.restype = ctypes.c_int32
    lib.gms_test_fixed_point.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    return lib


def _run(y, k):
    y = np.ascontiguousarray(y, np.float32); k = np.ascontiguousarray(k, np.int32)
    fixed = np.zeros(y.shape[0], np.int64); back = np.zeros(y.shape[0], np.float32)
    rc = _hook().gms_test_fixed_point(y.shape[0], y.ctypes.data, k.ctypes.data, fixed.ctypes.data, back.ctypes.data)
    assert rc == 0, rc
    return fixed, back


def _round_half_even(fr):
    n, d = fr.numerator, fr.denominator
    q, r = divmod(n, d)
    if 2 * r > d or (2 * r == d and q % 2 == 1):
        q += 1
    return q


def _exact(y, k):
    return np.array([_round_half_even(Fraction(float(a)) * Fraction(2) ** int(b)) for a, b in zip(y, k)], dtype=object)


def test_conversion_is_the_exact_rounding():
    rng = np.random.default_rng(5)
    n = 20000
    # magnitudes over the float range the kernel can meet; exponents chosen so that |y * 2^k| spans 2^-30 ... 2^47
    e_y = rng.integers(-60, 40, n)
    y = (rng.uniform(1.0, 2.0, n) * rng.choice([-1.0, 1.0], n)).astype(np.float32) * np.exp2(e_y).astype(np.float32)
    target = rng.integers(-30, 47, n)                       # exponent of the scaled value
    k = np.clip(target - e_y, -100, 100).astype(np.int32)
    fixed, back = _run(y, k)
    want = _exact(y, k)
    keep = np.array([abs(int(w)) < 2 ** 50 for w in want])
    assert keep.sum() > 0.95 * n
    assert all(int(a) == int(b) for a, b in zip(fixed[keep], want[keep]))
    # and back: the nearest float of fixed * 2^-k (exact in double for |fixed| < 2^53)
    ref = np.array([float(np.float32(np.ldexp(np.float64(int(f)), -int(kk)))) for f, kk in zip(fixed[keep], k[keep])], np.float32)
    assert np.array_equal(back[keep], ref)
    # values with at least 24 bits above the fixed-point unit come back unchanged
    exact_rep = keep & (e_y + k >= 24)                        # (the exponent of the scaled value actually used)
    assert exact_rep.sum() > 1000 and np.array_equal(back[exact_rep], y[exact_rep])


def test_edge_values():
    y = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 1.5, 2.5, -0.5, -1.5, 2.0 ** 47, -(2.0 ** 47), 1e-45, -1e-45, 3.0e38 * 0 + 2.0 ** -126, 123456.789], np.float32)
    k = np.array([10, 10, 0, 0, 0, 0, 0, 0, 0, 0, 0, 100, 100, 100, -3], np.int32)
    fixed, back = _run(y, k)
    want = _exact(y, k)
    assert [int(v) for v in fixed] == [int(v) for v in want]
    assert fixed[0] == 0 and fixed[1] == 0 and back[0] == 0.0          # a zero sum adds nothing
    assert [int(v) for v in fixed[4:9]] == [0, 2, 2, 0, -2]             # ties to even


def test_sixteen_adds_are_order_independent_and_cannot_overflow():
    rng = np.random.default_rng(9)
    y = (rng.uniform(-1.0, 1.0, (4000, 16)) * np.exp2(rng.integers(-20, 1, (4000, 16)))).astype(np.float32)       # |y| <= 1 = 2^0
    k = np.full(y.size, 47, np.int32)                                                                                # E = 0
    fixed, _ = _run(y.reshape(-1), k)
    fixed = fixed.reshape(4000, 16)
    assert np.abs(fixed).max() <= 2 ** 47
    s1 = fixed.sum(axis=1)
    s2 = fixed[:, rng.permutation(16)].sum(axis=1)
    assert np.array_equal(s1, s2) and np.abs(s1).max() < 2 ** 51
    # the fixed-point sum is the exact sum of the float inputs to within 16 half-units
    exact = y.astype(np.float64).sum(axis=1) * 2.0 ** 47
    assert np.abs(s1.astype(np.float64) - exact).max() <= 8.0 + 1e-6 * np.abs(exact).max()


def test_out_of_range_and_non_finite_values_saturate():
    """a partial sum beyond 2^50 fixed-point units -- a bound exceeded more than 8-fold, or a non-finite pixel
    gradient -- is clamped to +-2^50 instead of leaving arbitrary mantissa bits in the table: sixteen such adds still fit 64 bits, and
    every other entry of the unit keeps its own exact value."""
    y = np.array([2.0 ** 52, -(2.0 ** 52), 2.0 ** 60, np.inf, -np.inf, np.nan, 2.0 ** 50, 2.0 ** 49, -(2.0 ** 49), 1.0], np.float32)
    k = np.zeros(y.size, np.int32)
    fixed, back = _run(y, k)
    lim = 2 ** 50
    assert [int(v) for v in fixed[:5]] == [lim, -lim, lim, lim, -lim]
    assert abs(int(fixed[5])) == lim                               # NaN: a finite, saturated entry (v_med3 orders it below everything)
    assert [int(v) for v in fixed[6:]] == [lim, 2 ** 49, -(2 ** 49), 1]
    assert np.isfinite(back).all()
    assert 16 * lim < 2 ** 63                                       # sixteen saturated adds cannot wrap the 64-bit sum
