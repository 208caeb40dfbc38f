#!/usr/bin/env python3
"""Soak of the exact periodicity shortcut: random interior-rich views at high iteration caps,
shortcut on vs off (both on the device): final positions bit for bit, escape indices, bytes."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fractal_renderer_amd as fr  # noqa: E402
from fractal_renderer_amd import _native  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
fr.init(0)
lib = _native.load()
centres = [(-0.6, 0.0), (-1.0, 0.0), (-0.12, 0.75), (0.28, 0.53), (-1.7549, 0.0), (-0.75, 0.05), (0.3, 0.0), (-1.31, 0.0),
           (-0.7436447860, 0.1318252536), (-0.16, 1.035)]
bad = 0
t0 = time.time()
for seed in range(n):
    rng = np.random.default_rng(77_000 + seed)
    julia = rng.random() < 0.3
    cfg = fr.Config.new(fr.Algo.Julia if julia else fr.Algo.Mandelbrot)
    cfg.width, cfg.height = int(rng.integers(100, 700)), int(rng.integers(64, 500))
    cfg.iterations = int(rng.choice([257, 1000, 1024, 2047, 4096, 10000, 33333]))
    cfg.exposure = 5.0
    cx, cy = centres[int(rng.integers(len(centres)))]
    sc = float(10 ** rng.uniform(-0.5, 3.5))
    cfg.scale.re = cfg.scale.im = sc
    cfg.pos.re, cfg.pos.im = cx + float(rng.normal(0, 0.2 / sc)), cy + float(rng.normal(0, 0.2 / sc))
    if julia:
        # parameters inside the Mandelbrot set give filled Julia sets (attracting cycles everywhere inside)
        jc = [(-0.123, 0.745), (-1.0, 0.05), (0.25, 0.0), (-0.5, 0.5), (0.0, 0.6), (-0.8, 0.156)][int(rng.integers(6))]
        cfg.julia_set.re, cfg.julia_set.im = jc
        cfg.pos.re, cfg.pos.im = float(rng.normal(0, 0.3)), float(rng.normal(0, 0.3))
        cfg.scale.re = cfg.scale.im = float(10 ** rng.uniform(-0.5, 1.5))
    cfg.limit = float(rng.choice([65536.0, 65536.0, 100.0, 4.0, 1000.5]))
    cfg.smooth = int(rng.random() < 0.8)
    prec = fr.Precision.F32 if rng.random() < 0.3 else fr.Precision.F64
    lib.fr_set_tile(int(rng.choice([0, 9, 9])))
    lib.fr_set_loop_mode(int(rng.choice([-1, -1, 2, 4])))
    lib.fr_set_cycle_shortcut(0)
    z0, it0 = fr.escape_rows(cfg, precision=prec)
    img0 = fr.get_image(cfg, prec)
    lib.fr_set_cycle_shortcut(1)
    z1, it1 = fr.escape_rows(cfg, precision=prec)
    img1 = fr.get_image(cfg, prec)
    ok = np.array_equal(it0, it1) and np.array_equal(z0.view(np.uint64), z1.view(np.uint64)) and np.array_equal(img0, img1)
    if not ok:
        bad += 1
        print("MISMATCH", seed, bytes(cfg).hex(), int(prec), flush=True)
    if seed % 50 == 49:
        capped = float((it0 == cfg.iterations).mean())
        print("%d configs, %d mismatches, %.0f s (last: %dx%d it=%d, %.0f %% at the cap)" % (
            seed + 1, bad, time.time() - t0, cfg.width, cfg.height, cfg.iterations, 100 * capped), flush=True)
lib.fr_set_cycle_shortcut(0); lib.fr_set_tile(0); lib.fr_set_loop_mode(-1)
print("done: %d configs, %d mismatches" % (n, bad))
sys.exit(1 if bad else 0)
