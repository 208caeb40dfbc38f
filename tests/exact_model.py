"""calc::recursive (calc/src/lib.rs:245-257) in EXACT rational arithmetic with explicit IEEE-754
round-to-nearest-even after every operation the reference performs — independent of any compiler,
libm or FPU: `fractions.Fraction` for the exact value, integer arithmetic for the rounding.

Used to pin the "no FMA" rule: Rust never contracts a*b+c, so

    re' = fl(fl(fl(re*re) - fl(im*im)) + c.re)       (square :88, Add :103)
    im' = fl(fl(fl(2.0*re) * im) + c.im)             (square :89, Add :104)
    dist = fl(fl(re'*re') + fl(im'*im'))             (squared_distance :95)

and a compiler that fuses any multiply-add pair changes low bits.  `recursive(..., fuse=...)` also
models those contractions, so a test can show that its inputs are sensitive to each of them.
"""
from fractions import Fraction

FORMATS = {"f64": (53, -1022, 1023), "f32": (24, -126, 127)}


def rne(x, fmt="f64"):
    """Round the exact rational x to the nearest value of the binary format (ties to even); the result
    is returned as a Fraction.  Overflow is not modelled (raises): the KATs stay in range."""
    if x == 0:
        return Fraction(0)
    p, emin, emax = FORMATS[fmt]
    sign = -1 if x < 0 else 1
    a = -x if x < 0 else x
    # e = floor(log2(a)), from the bit lengths of numerator and denominator
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fraction(2) ** e > a:
        e -= 1
    elif Fraction(2) ** (e + 1) <= a:
        e += 1
    assert Fraction(2) ** e <= a < Fraction(2) ** (e + 1)
    q = max(e, emin) - (p - 1)  # exponent of the unit in the last place (subnormals share emin's)
    scaled = a / Fraction(2) ** q  # the significand as a rational; round it to an integer, ties to even
    n = scaled.numerator // scaled.denominator
    rem = scaled - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and n % 2 == 1):
        n += 1
    r = Fraction(n) * Fraction(2) ** q
    if r >= Fraction(2) ** (emax + 1):
        raise OverflowError("result overflows the format")
    return sign * r


def to_float(x):
    """Exact Fraction (already a binary64 value) -> Python float."""
    f = x.numerator / x.denominator  # CPython's int/int is correctly rounded; x is representable, so exact
    assert Fraction(f) == x
    return f


def recursive(iterations, start, c, limit, fmt="f64", fuse=None):
    """The reference loop.  start, c: pairs of floats (exact binary values); returns ((re, im), index).
    fuse: None = the reference's roundings; otherwise one contraction a fusing compiler could apply:
      'sq_re_a'  fma(re, re, -fl(im*im))          'sq_re_b'  fma(-im, im, fl(re*re))
      'add_im'   fma(fl(2re), im, c.im)            'dist_a'   fma(re', re', fl(im'*im'))
      'dist_b'   fma(im', im', fl(re'*re'))        'add_re'   (no product feeds this add directly: a
                                                              compiler cannot fuse it; listed for clarity)
    """
    R = lambda v: rne(v, fmt)  # noqa: E731
    lim = R(Fraction(limit))
    squared = R(lim * lim)
    pre, pim = R(Fraction(start[0])), R(Fraction(start[1]))
    cre, cim = R(Fraction(c[0])), R(Fraction(c[1]))
    for i in range(iterations):
        if fuse == "sq_re_a":
            sq_re = R(pre * pre - R(pim * pim))
        elif fuse == "sq_re_b":
            sq_re = R(R(pre * pre) - pim * pim)
        else:
            sq_re = R(R(pre * pre) - R(pim * pim))
        two_re = R(2 * pre)
        if fuse == "add_im":
            nim = R(two_re * pim + cim)
        else:
            nim = R(R(two_re * pim) + cim)
        nre = R(sq_re + cre)
        if fuse == "dist_a":
            dist = R(nre * nre + R(nim * nim))
        elif fuse == "dist_b":
            dist = R(R(nre * nre) + nim * nim)
        else:
            dist = R(R(nre * nre) + R(nim * nim))
        if dist > squared:
            return (to_float(nre), to_float(nim)), i
        pre, pim = nre, nim
    return (to_float(pre), to_float(pim)), iterations


FUSIONS = ["sq_re_a", "sq_re_b", "add_im", "dist_a", "dist_b"]

# Contraction-sensitive known-answer inputs: (iterations, start == c (Mandelbrot), limit).  Found by
# tests/golden/find_contraction_kats.py (seeded search); each changes (final position bits or escape
# index) under EVERY orbit-affecting contraction within the given iteration count, and the two
# dist contractions flip the escape decision of the last one.
CONTRACTION_KATS_F64 = [
    (3, float.fromhex('-0x1.8449ad7c7cde2p-1'), float.fromhex('-0x1.1ea6ced0b8d10p-1'), float.fromhex('0x1.0000000000000p+16')),
    (4, float.fromhex('-0x1.24626187c209ep+0'), float.fromhex('0x1.60628cdd0fcf0p-1'), float.fromhex('0x1.0000000000000p+16')),
    (4, float.fromhex('0x1.b5627f0bcfb40p-5'), float.fromhex('0x1.96edcf4c448c8p-1'), float.fromhex('0x1.0000000000000p+16')),
    (3, float.fromhex('-0x1.f7ab1302cee70p-3'), float.fromhex('-0x1.e9a4f08d5d48ep-1'), float.fromhex('0x1.0000000000000p+16')),
    (5, float.fromhex('-0x1.5d36f3cba7977p+0'), float.fromhex('-0x1.79bccc52615a8p-1'), float.fromhex('0x1.0000000000000p+16')),
    (5, float.fromhex('-0x1.023dbae99bbafp+0'), float.fromhex('0x1.ef5c1bb27d08ep-1'), float.fromhex('0x1.0000000000000p+16')),
    (5, float.fromhex('-0x1.2a89cea652202p+0'), float.fromhex('-0x1.93b195ef210c0p-1'), float.fromhex('0x1.90d4bcf00ac0bp+2')),
    (5, float.fromhex('-0x1.58f6ca312a0cfp+0'), float.fromhex('0x1.ae920ff68778ap-1'), float.fromhex('0x1.7b1d111f0f27cp+3')),
    (4, float.fromhex('0x1.5714f2b9bb294p-2'), float.fromhex('-0x1.36245c2b64dc4p-2'), float.fromhex('0x1.63325a7a17ef7p-1')),
    (3, float.fromhex('-0x1.5b8f5f823d4b8p-1'), float.fromhex('-0x1.e1dd16176b23cp-2'), float.fromhex('0x1.e1e055f5b29c4p-2')),
]
CONTRACTION_KATS_F32 = [
    (3, float.fromhex('-0x1.35e0980000000p+0'), float.fromhex('0x1.5660480000000p-1'), float.fromhex('0x1.0000000000000p+16')),
    (5, float.fromhex('-0x1.656b260000000p-2'), float.fromhex('-0x1.7afbd80000000p-1'), float.fromhex('0x1.0000000000000p+16')),
    (5, float.fromhex('0x1.42e9560000000p-3'), float.fromhex('0x1.6063e40000000p-3'), float.fromhex('0x1.0000000000000p+16')),
    (5, float.fromhex('-0x1.7a83120000000p-1'), float.fromhex('-0x1.3d91f20000000p-1'), float.fromhex('0x1.0000000000000p+16')),
    (4, float.fromhex('-0x1.a9204a0000000p-1'), float.fromhex('-0x1.e7e63e0000000p-2'), float.fromhex('0x1.105c380000000p+0')),
    (4, float.fromhex('-0x1.cebb920000000p-1'), float.fromhex('-0x1.d28d900000000p-1'), float.fromhex('0x1.2b95e60000000p+1')),
    (3, float.fromhex('0x1.c263080000000p-3'), float.fromhex('0x1.fe59060000000p-1'), float.fromhex('0x1.9ba84e0000000p+0')),
]
