#!/usr/bin/env python3
"""Seeded search for CONTRACTION-SENSITIVE known-answer inputs of calc::recursive
(calc/src/lib.rs:245-257), using only tests/exact_model.py (exact rationals + explicit rounding).

Prints Python literals to paste into exact_model.CONTRACTION_KATS_F64 / _F32:
  * orbit KATs: start == c (Mandelbrot), whose final position after `n` iterations changes under EACH
    of the three orbit-affecting contractions (sq_re_a, sq_re_b, add_im);
  * escape-decision KATs: an orbit and a limit with fl(limit*limit) == fl(re'^2 + im'^2) exactly at
    iteration k (so the reference does NOT escape there: `>` is strict) while fma(re', re', fl(im'^2))
    or fma(im', im', fl(re'^2)) is one ulp larger (a contracting compiler escapes one iteration early).
"""
import math
import os
import random
import struct
import sys
from fractions import Fraction

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import exact_model as M  # noqa: E402


def f32(x):
    return struct.unpack("f", struct.pack("f", x))[0]


def pixel_exact(v):
    """Can a 1x1 image put a pixel exactly at v?  Its coordinate is fl(-0.5 / 1.0 + pos) with pos = v + 0.5
    (calc/src/lib.rs:182-197, width = height = 1, scale = 1): both steps must be exact."""
    pos = Fraction(v) + Fraction(1, 2)
    return M.rne(pos, "f64") == pos


def orbit_kats(fmt, count, rng):
    out = []
    while len(out) < count:
        re, im = rng.uniform(-1.6, 0.4), rng.uniform(-1.0, 1.0)
        if fmt == "f32":
            re, im = f32(re), f32(im)
        if not (pixel_exact(re) and pixel_exact(im)):
            continue
        n = rng.choice([3, 4, 5])
        base = M.recursive(n, (re, im), (re, im), 65536.0, fmt)
        if base[1] != n:
            continue  # keep it bounded for n iterations so every iterate is exercised
        if all(M.recursive(n, (re, im), (re, im), 65536.0, fmt, f) != base for f in ["sq_re_a", "sq_re_b", "add_im"]):
            out.append((n, re, im, 65536.0))
    return out


def escape_kats(fmt, count, rng):
    out = []
    tries = 0
    while len(out) < count:
        tries += 1
        re, im = rng.uniform(-1.6, 0.4), rng.uniform(-1.0, 1.0)
        if fmt == "f32":
            re, im = f32(re), f32(im)
        if not (pixel_exact(re) and pixel_exact(im)):
            continue
        k = rng.choice([1, 2, 3])
        pos, it = M.recursive(k, (re, im), (re, im), 1e15 if fmt == "f32" else 1e150, fmt)
        if it != k:
            continue
        R = lambda v: M.rne(v, fmt)  # noqa: E731
        nre, nim = Fraction(pos[0]), Fraction(pos[1])
        d = R(R(nre * nre) + R(nim * nim))
        da = R(nre * nre + R(nim * nim))
        db = R(R(nre * nre) + nim * nim)
        if not (da > d or db > d):
            continue
        # a limit whose rounded square is exactly d
        lim0 = math.sqrt(float(d))
        cand = lim0
        found = None
        for step in range(-8, 9):
            c = lim0
            for _ in range(abs(step)):
                c = math.nextafter(c, math.inf if step > 0 else 0.0)
            if fmt == "f32":
                c = f32(c)
            if R(R(Fraction(c)) * R(Fraction(c))) == d:
                found = c
                break
        if found is None:
            continue
        # iteration index k-1 produced `pos` (k iterations done): with limit `found` the reference runs on
        base = M.recursive(k + 2, (re, im), (re, im), found, fmt)
        fa = M.recursive(k + 2, (re, im), (re, im), found, fmt, "dist_a")
        fb = M.recursive(k + 2, (re, im), (re, im), found, fmt, "dist_b")
        if base != fa or base != fb:
            out.append((k + 2, re, im, found))
    return out


def show(name, rows):
    print("%s = [" % name)
    for n, re, im, lim in rows:
        print("    (%d, float.fromhex(%r), float.fromhex(%r), float.fromhex(%r))," % (n, re.hex(), im.hex(), float(lim).hex()))
    print("]")


if __name__ == "__main__":
    rng = random.Random(20261004)
    show("CONTRACTION_KATS_F64", orbit_kats("f64", 6, rng) + escape_kats("f64", 4, rng))
    show("CONTRACTION_KATS_F32", orbit_kats("f32", 4, rng) + escape_kats("f32", 3, rng))
