"""Pins that do not depend on any compiler, libm or FPU (exact rationals + explicit rounding,
tests/exact_model.py), for the two places where the oracle could silently drift from the reference:

  1. "no FMA" (SURVEY.md fact 3): contraction-SENSITIVE known-answer tests of calc::recursive
     (calc/src/lib.rs:88-89, 95, 103-104) — inputs whose results change under every contraction a
     fusing compiler could apply.  The C oracle (f64 and the build-defined f32 form) and the independent
     pure-Python restatement must reproduce the exactly-rounded model; the fused variants must not.
  2. the software log2 that stands in for f64::log2 (calc/src/lib.rs:222-223; fr_math.h, shared by the
     kernels and the oracle's SOFT mode): checked against a 60-digit `decimal` log2 on the inputs the
     colour map actually produces, so that the shared source is itself pinned (error < 0.53 ulp, and
     it never differs from glibc's by more than one ulp).
"""
import decimal
import math
import random
import struct
from fractions import Fraction

import numpy as np
import pytest

import exact_model as M
import oracle_lib as O
from test_oracle_numpy_cross import recursive as py_recursive


def bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


# ---- the rounding model itself -----------------------------------------------------------------


def test_rne_is_ieee_round_to_nearest_even():
    rng = random.Random(7)
    for _ in range(3000):
        num = rng.getrandbits(rng.choice([10, 60, 120, 200])) + 1
        den = rng.getrandbits(rng.choice([1, 30, 90, 150])) + 1
        f = Fraction(num, den) * Fraction(2) ** rng.randint(-80, 80) * rng.choice([-1, 1])
        assert M.to_float(M.rne(f)) == f.numerator / f.denominator  # CPython int/int: correctly rounded
    # ties go to even; subnormals; f32 against numpy on products of two f32 values (exact in f64)
    assert M.rne(Fraction(2 ** 53 + 1)) == 2 ** 53 and M.rne(Fraction(2 ** 53 + 3)) == 2 ** 53 + 4
    assert M.rne(Fraction(1, 2 ** 1075)) == 0 and M.rne(Fraction(3, 2 ** 1075)) == Fraction(1, 2 ** 1073)
    for _ in range(2000):
        a = float(np.float32(rng.uniform(-4, 4)))
        b = float(np.float32(rng.uniform(-4, 4)))
        assert float(np.float32(a * b)) == M.to_float(M.rne(Fraction(a) * Fraction(b), "f32"))
        assert float(np.float32(a) + np.float32(b)) == M.to_float(M.rne(Fraction(a) + Fraction(b), "f32"))


# ---- 1. contraction-sensitive KATs ------------------------------------------------------------------


@pytest.mark.parametrize("kat", M.CONTRACTION_KATS_F64)
def test_contraction_sensitive_kats_f64(kat):
    n, re, im, limit = kat
    want = M.recursive(n, (re, im), (re, im), limit)
    # the C oracle (gcc -O2 -ffp-contract=off) and the independent Python restatement
    (ore, oim), oit = O.recursive(n, (re, im), (re, im), limit)
    assert (bits(ore), bits(oim), oit) == (bits(want[0][0]), bits(want[0][1]), want[1])
    (pre, pim), pit = py_recursive(n, (re, im), (re, im), limit)
    assert (bits(pre), bits(pim), pit) == (bits(want[0][0]), bits(want[0][1]), want[1])
    # ... and the KAT really is sensitive: some contraction changes it
    changed = [f for f in M.FUSIONS if M.recursive(n, (re, im), (re, im), limit, fuse=f) != want]
    if limit == 65536.0:
        assert {"sq_re_a", "sq_re_b", "add_im"} <= set(changed), changed
    else:
        assert {"dist_a", "dist_b"} & set(changed), changed


@pytest.mark.parametrize("kat", M.CONTRACTION_KATS_F32)
def test_contraction_sensitive_kats_f32(kat):
    n, re, im, limit = kat
    want = M.recursive(n, (re, im), (re, im), limit, "f32")
    (ore, oim), oit = O.recursive(n, (re, im), (re, im), limit, f32=True)
    assert (bits(ore), bits(oim), oit) == (bits(want[0][0]), bits(want[0][1]), want[1])
    changed = [f for f in M.FUSIONS if M.recursive(n, (re, im), (re, im), limit, "f32", fuse=f) != want]
    if limit == 65536.0:
        assert {"sq_re_a", "sq_re_b", "add_im"} <= set(changed), changed
    else:
        assert {"dist_a", "dist_b"} & set(changed), changed


def test_hand_kats_agree_with_the_exact_model():
    """SURVEY.md §8c KAT-1..3 through the exact model (sanity of the model against hand arithmetic)."""
    assert M.recursive(50, (2.0, 0.0), (2.0, 0.0), 65536.0) == ((2090918.0, 0.0), 3)
    assert M.recursive(50, (2.0, 0.0), (2.0, 0.0), 2.0) == ((6.0, 0.0), 0)
    assert M.recursive(50, (-2.0, 0.0), (-2.0, 0.0), 65536.0) == ((2.0, 0.0), 50)


def test_exact_model_tracks_the_oracle_on_random_orbits():
    rng = random.Random(11)
    for _ in range(60):
        s = (rng.uniform(-2, 1), rng.uniform(-1.5, 1.5))
        c = s if rng.random() < 0.6 else (rng.uniform(-1, 1), rng.uniform(-1, 1))
        n = rng.randint(1, 12)
        lim = rng.choice([65536.0, 2.0, 3.7])
        assert M.recursive(n, s, c, lim) == O.recursive(n, s, c, lim)
        s32 = tuple(float(np.float32(v)) for v in s)
        c32 = tuple(float(np.float32(v)) for v in c)
        assert M.recursive(n, s32, c32, lim, "f32") == O.recursive(n, s32, c32, lim, f32=True)


# ---- 2. the software log2 against a high-precision log2 --------------------------------------------


def _ulp(x):
    return math.ulp(x)


def _exact_log2(x, ctx):
    return ctx.divide(ctx.ln(decimal.Decimal(x)), ctx.ln(decimal.Decimal(2)))


def colour_path_log2_inputs(rng, n):
    """What calc/src/lib.rs:222-223 feeds to log2: sqrt(dist) with dist just past limit^2 (2^32 .. ~2^66
    at the CLI's limit), other limits, then log_zn = log2(sqrt(dist)) / 2."""
    xs = []
    for _ in range(n):
        kind = rng.random()
        if kind < 0.45:
            dist = 2.0 ** rng.uniform(32, 66)
        elif kind < 0.6:
            dist = 2.0 ** rng.uniform(1.001, 32)
        elif kind < 0.7:
            dist = 2.0 ** rng.uniform(66, 900)
        else:
            dist = None
        if dist is not None:
            xs.append(math.sqrt(dist))
        else:
            xs.append(rng.uniform(0.25, 300.0) if rng.random() < 0.8 else 1.0 + rng.uniform(-0.06, 0.06))
    return xs


def test_soft_log2_error_bound_against_60_digit_log2():
    ctx = decimal.Context(prec=60)
    rng = random.Random(2026)
    xs = colour_path_log2_inputs(rng, 6000) + [2.0 ** k for k in range(-20, 80, 7)] + [0.96875, 1.03125, 1.0 - 2 ** -53]
    O.set_log2_mode(O.LOG2_SOFT)
    try:
        soft = [O.log2(x) for x in xs]
    finally:
        O.set_log2_mode(O.LOG2_LIBM)
    worst, misrounded, vs_libm = 0.0, 0, 0
    for x, s in zip(xs, soft):
        exact = _exact_log2(x, ctx)
        if exact == 0:
            assert s == 0.0
            continue
        # error of the returned double in ulps of the correctly rounded result
        cr = float(exact)
        err = abs((decimal.Decimal(s) - exact) / decimal.Decimal(_ulp(cr)))
        worst = max(worst, float(err))
        misrounded += s != cr
        vs_libm += abs(bits(s) - bits(math.log2(x))) if (s > 0) == (math.log2(x) > 0) else 99
        assert abs(bits(s) - bits(math.log2(x))) <= 1 or s == math.log2(x)
    assert worst < 0.53, worst
    assert misrounded <= len(xs) * 0.02, (misrounded, len(xs))  # it almost always IS the correctly rounded value


def test_libm_and_soft_colour_bytes_agree_on_golden_cases(oracle):
    """The byte a pixel gets does not depend on which of the two log2s is used, on any golden case: the
    only place the reference delegates to the platform (f64::log2) does not reach the output here."""
    import golden_util as G

    for key in G.KEYS:
        cfg, prec = G.oracle_config(key), G.precision_of(key)
        O.set_log2_mode(O.LOG2_SOFT)
        try:
            soft = O.get_image(cfg, prec)
        finally:
            O.set_log2_mode(O.LOG2_LIBM)
        assert np.array_equal(soft, O.get_image(cfg, prec)), key
