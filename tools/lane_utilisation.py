#!/usr/bin/env python3
"""Useful-lane fraction of the one-lane-per-pixel mapping: Σ executed iterations ÷
Σ over 8x8 wave tiles of 64 x (max executed iterations in the tile), from escape indices dumped by
the device (fr_escape_rows) for the BASELINE views.  Runs on the GPU box."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fractal_renderer_amd as fr  # noqa: E402

fr.init(0)
VIEWS = {
    "C2 default i=1024 f64": dict(it=1024, prec=fr.Precision.F64),
    "C3 zoom1e6 i=65536 f64": dict(it=65536, prec=fr.Precision.F64, pos=(-0.7436447860, 0.1318252536), scale=1e6),
    "C4 julia i=4096 f32": dict(it=4096, prec=fr.Precision.F32, julia=(-0.8, 0.156)),
}
N = 16384
for name, v in VIEWS.items():
    cfg = fr.Config.new(fr.Algo.Julia if "julia" in v else fr.Algo.Mandelbrot)
    cfg.width = cfg.height = N
    cfg.iterations, cfg.exposure = v["it"], 5.0
    if "julia" in v:
        cfg.julia_set.re, cfg.julia_set.im = v["julia"]
    else:
        cfg.pos.re, cfg.pos.im = v.get("pos", (-0.6, 0.0))
    if "scale" in v:
        cfg.scale.re = cfg.scale.im = v["scale"]
    tot = 0
    cost = {(8, 8): 0, (16, 4): 0, (64, 1): 0}
    hist = np.zeros(65, dtype=np.int64)  # waves by number of lanes still running at 1/8 of the wave's life
    for y0 in range(0, N, 1024):
        it = np.empty((1024, N), dtype=np.uint32)
        import ctypes as C
        from fractal_renderer_amd import _native
        _native.check(_native.load().fr_escape_rows(C.byref(cfg), int(v["prec"]), y0, y0 + 1024, None, it.ctypes.data))
        ex = np.where(it < cfg.iterations, it.astype(np.int64) + 1, cfg.iterations)
        tot += int(ex.sum())
        for (tw, th) in cost:
            m = ex.reshape(1024 // th, th, N // tw, tw).max(axis=(1, 3))
            cost[(tw, th)] += int(m.sum()) * 64
    print("%-26s Σ executed %.4e   useful-lane fraction: %s" % (
        name, tot, ", ".join("%dx%d %.3f" % (k[0], k[1], tot / c) for k, c in cost.items())), flush=True)
