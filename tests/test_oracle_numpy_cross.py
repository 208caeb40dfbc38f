"""A second, independent restatement of the reference path in pure Python (float = IEEE binary64,
every operation separately rounded, math.log2 = the platform libm like Rust's f64::log2), written
straight from calc/src/lib.rs without looking at oracle/fractal_oracle.c, cross-checked against the C
oracle on seeded random inputs.  Two independent readings of the source agreeing bit for bit is the
strongest pin available while the reference itself cannot be built (no Rust toolchain)."""
import math

import numpy as np
import pytest

import oracle_lib as O


def rust_as_u8(v):  # `f64 as u8`: saturating, truncating, NaN -> 0
    if v != v:
        return 0
    if v <= 0.0:
        return 0
    if v >= 255.0:
        return 255
    return int(v)


def rgb_new(r, b, g):  # calc/src/lib.rs:129-131 — parameters (r, b, g), fields {r, g, b}
    return {"r": r, "g": g, "b": b}


def color_multiply(color, mult):  # :133-139
    return rgb_new(rust_as_u8(float(color["r"]) * mult), rust_as_u8(float(color["g"]) * mult),
                   rust_as_u8(float(color["b"]) * mult))


def recursive(iterations, start, c, limit):  # :245-257
    squared = limit * limit
    prev_re, prev_im = start
    for i in range(iterations):
        sq_re = (prev_re * prev_re) - (prev_im * prev_im)  # Imaginary::square :87-92
        sq_im = 2.0 * prev_re * prev_im
        next_re = sq_re + c[0]  # Add :98-107
        next_im = sq_im + c[1]
        dist = next_re * next_re + next_im * next_im  # squared_distance :94-96
        if dist > squared:
            return (next_re, next_im), i
        prev_re, prev_im = next_re, next_im
    return (prev_re, prev_im), iterations


def coord_to_space(coord, mx, offset, pos, scale):  # :182-184
    return ((coord / mx) - offset) / scale + pos


def get_recursive_pixel(cfg, x, y):  # :199-235
    width, height = float(cfg.width), float(cfg.height)
    start = (coord_to_space(float(x), height, (width / height) / 2.0, cfg.pos.re, cfg.scale.re),
             coord_to_space(float(y), height, 0.5, cfg.pos.im, cfg.scale.im))
    if cfg.algo == O.MANDELBROT:
        pos, iters = recursive(cfg.iterations, start, start, cfg.limit)
    elif cfg.algo == O.JULIA:
        pos, iters = recursive(cfg.iterations, start, (cfg.julia_set.re, cfg.julia_set.im), cfg.limit)
    else:
        return (0, 0, 0), None, None
    dist = pos[0] * pos[0] + pos[1] * pos[1]
    primary = {"r": cfg.primary_color.r, "g": cfg.primary_color.g, "b": cfg.primary_color.b}
    secondary = {"r": cfg.secondary_color.r, "g": cfg.secondary_color.g, "b": cfg.secondary_color.b}
    if dist > cfg.stable_limit:
        it = float(iters)
        if cfg.smooth:
            def log2(v):  # f64::log2 with IEEE semantics at the edges (math.log2 raises instead)
                if v != v or v < 0.0:
                    return math.nan
                if v == 0.0:
                    return -math.inf
                return math.log2(v)
            log_zn = log2(math.sqrt(dist)) / 2.0
            nu = log2(log_zn)
            it += 1.0 - nu
        denom = float(cfg.iterations)
        if denom == 0.0:  # IEEE division by zero, which Python refuses to do
            q = math.nan if (it == 0.0 or it != it) else math.copysign(math.inf, it)
        else:
            q = it / denom
        col = color_multiply(primary, q * cfg.exposure)
    elif cfg.inside:
        col = color_multiply(secondary, dist)
    else:
        col = rgb_new(0, 0, 0)
    return (col["r"], col["g"], col["b"]), pos, iters


def _random_cfg(rng):
    algo = O.JULIA if rng.random() < 0.4 else O.MANDELBROT
    kw = dict(iterations=int(rng.choice([0, 1, 3, 17, 50, 120])),
              pos=(float(rng.normal(-0.5, 0.6)), float(rng.normal(0, 0.6))),
              scale=(float(10 ** rng.uniform(-0.7, 3)), float(10 ** rng.uniform(-0.7, 3))),
              limit=float(rng.choice([65536.0, 2.0, 100.0, 0.5, 1e200])), stable_limit=float(rng.choice([2.0, 0.5, 0.0, 30.0])),
              exposure=float(rng.choice([5.0, 2.0, 50.0, -1.0])), inside=int(rng.random() < 0.7), smooth=int(rng.random() < 0.7),
              primary_color=tuple(int(v) for v in rng.integers(0, 256, 3)),
              secondary_color=tuple(int(v) for v in rng.integers(0, 256, 3)))
    if algo == O.JULIA:
        kw["julia_set"] = (float(rng.uniform(-1.2, 0.6)), float(rng.uniform(-0.8, 0.8)))
    return O.cli_config(int(rng.integers(1, 24)), int(rng.integers(1, 20)), algo, **kw)


@pytest.mark.parametrize("seed", range(25))
def test_c_oracle_agrees_with_independent_python_restatement(seed):
    rng = np.random.default_rng(2024 + seed)
    O.set_log2_mode(O.LOG2_LIBM)
    cfg = _random_cfg(rng)
    img = O.get_image(cfg)
    z, it = O.escape_rows(cfg)
    for y in range(cfg.height):
        for x in range(cfg.width):
            col, pos, iters = get_recursive_pixel(cfg, x, y)
            assert tuple(img[y, x]) == col, (seed, x, y)
            assert it[y, x] == iters
            for a, b in zip(z[y, x], pos):
                assert (a == b) or (a != a and b != b), (seed, x, y, a, b)


def test_python_restatement_passes_the_kats():
    assert recursive(50, (2.0, 0.0), (2.0, 0.0), 65536.0) == ((2090918.0, 0.0), 3)
    assert recursive(50, (-2.0, 0.0), (-2.0, 0.0), 65536.0) == ((2.0, 0.0), 50)
    cfg = O.config_new(width=4, height=4, scale=(0.25, 0.25))
    assert get_recursive_pixel(cfg, 0, 2)[0] == (83, 83, 255)
    assert get_recursive_pixel(cfg, 1, 2)[0] == (240, 170, 0)
