#!/usr/bin/env python3
"""Two-pass render (tile 11) against the default dispatch: bytes on a handful of configs (list overflow
included), then C4 timings over first_cap.  Runs on the GPU box."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch  # noqa: F401  (before the library)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fractal_renderer_amd as fr  # noqa: E402
from fractal_renderer_amd import _native  # noqa: E402

fr.init(0)
lib = _native.load()


def cfg_of(algo, w, h, it, julia=None, pos=None, scale=None, smooth=True, inside=True):
    cfg = fr.Config.new(algo)
    cfg.width, cfg.height, cfg.iterations = w, h, it
    if julia:
        cfg.julia_set.re, cfg.julia_set.im = julia
        cfg.pos.re = 0.0
    if pos:
        cfg.pos.re, cfg.pos.im = pos
    if scale:
        cfg.scale.re = cfg.scale.im = scale
    cfg.smooth, cfg.inside = smooth, inside
    return cfg


def image(cfg, prec, tile, minrun=-1):
    lib.fr_set_tile(tile)
    lib.fr_set_refill_policy(minrun, -1)
    img = fr.get_image(cfg, prec)
    lib.fr_set_tile(0)
    lib.fr_set_refill_policy(-1, -1)
    return img


bad = 0
cases = [
    ("julia 1500x1100 i=4096", cfg_of(fr.Algo.Julia, 1500, 1100, 4096, julia=(-0.8, 0.156))),
    ("julia 1003x777 i=300 unsmooth", cfg_of(fr.Algo.Julia, 1003, 777, 300, julia=(-0.8, 0.156), smooth=False)),
    ("julia interior 900x900 i=1000", cfg_of(fr.Algo.Julia, 900, 900, 1000, julia=(-0.12, 0.74))),
    ("mandelbrot 1200x900 i=1024", cfg_of(fr.Algo.Mandelbrot, 1200, 900, 1024)),
    ("mandelbrot zoom 1000x1000 i=5000", cfg_of(fr.Algo.Mandelbrot, 1000, 1000, 5000, pos=(-0.7436447860, 0.1318252536), scale=1e4)),
    ("julia 64x64 i=200", cfg_of(fr.Algo.Julia, 64, 64, 200, julia=(-0.8, 0.156))),
    ("julia 700x500 i=140 (cap just past first_cap)", cfg_of(fr.Algo.Julia, 700, 500, 140, julia=(-0.8, 0.156))),
]
for name, cfg in cases:
    for prec in (fr.Precision.F32, fr.Precision.F64):
        want = image(cfg, prec, 0)
        for forced in (0, 512, 0):
            lib.fr_debug_set_two_pass_capacity(forced)
            for k1 in (-1, 16, 64):
                got = image(cfg, prec, 11, k1)
                ok = np.array_equal(got, want)
                if not ok:
                    bad += 1
                    d = np.argwhere((got != want).any(axis=2))
                    print("MISMATCH", name, prec, "capacity", forced, "first_cap", k1, len(d), "px; first", d[0], got[d[0][0], d[0][1]], want[d[0][0], d[0][1]], flush=True)
        lib.fr_debug_set_two_pass_capacity(0)
        print("ok " if not bad else "BAD", name, prec, flush=True)
if bad:
    sys.exit(1)

# C4 timings
import torch
for prec, pn in ((fr.Precision.F32, "f32"), (fr.Precision.F64, "f64")):
    cfg = cfg_of(fr.Algo.Julia, 16384, 16384, 4096, julia=(-0.8, 0.156))
    out = torch.empty(16384 * 16384 * 3, dtype=torch.uint8, device="cuda")
    ref = None
    for tile, k1 in ((0, -1), (10, -1), (11, 64), (11, 96), (11, 128), (11, 160), (11, 192), (11, 256)):
        lib.fr_set_tile(tile)
        lib.fr_set_refill_policy(k1, -1)
        ts = []
        for rep in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _native.check(lib.fr_render_rows_rgb8_device(C.byref(cfg), int(prec), 0, 16384, C.c_void_p(out.data_ptr()), out.numel(), None))
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        kn = C.create_string_buffer(256)
        lib.fr_last_kernel_name(kn, 256)
        if ref is None:
            ref = out.clone()
        same = bool(torch.equal(out, ref))
        print("C4 %s tile %2d first_cap %4d: best %.3f ms  median %.3f  identical to default: %s  [%s]" % (pn, tile, k1, min(ts), sorted(ts)[2], same, kn.value.decode()[:60]), flush=True)
    lib.fr_set_tile(0)
    lib.fr_set_refill_policy(-1, -1)
