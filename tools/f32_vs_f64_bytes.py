#!/usr/bin/env python3
"""SURVEY.md §8a note: fraction of output bytes on which the f32 fast path differs from the f64
(reference) arithmetic, for the BASELINE views at 4096^2 (runs on the GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fractal_renderer_amd as fr  # noqa: E402

fr.init(0)
N = 4096
views = {
    "C2 default i=1024": dict(it=1024),
    "C4 julia i=4096": dict(it=4096, julia=(-0.8, 0.156)),
    "zoom 1e3 i=1024": dict(it=1024, pos=(-0.7436447860, 0.1318252536), scale=1e3),
    "zoom 1e6 i=1024 (C1/C3 view)": dict(it=1024, pos=(-0.7436447860, 0.1318252536), scale=1e6),
}
for name, v in views.items():
    cfg = fr.Config.new(fr.Algo.Julia if "julia" in v else fr.Algo.Mandelbrot)
    cfg.width = cfg.height = N
    cfg.iterations, cfg.exposure = v["it"], 5.0
    if "julia" in v:
        cfg.julia_set.re, cfg.julia_set.im = v["julia"]
    else:
        cfg.pos.re, cfg.pos.im = v.get("pos", (-0.6, 0.0))
    if "scale" in v:
        cfg.scale.re = cfg.scale.im = v["scale"]
    a = fr.get_image(cfg, fr.Precision.F64)
    b = fr.get_image(cfg, fr.Precision.F32)
    d = a != b
    big = np.abs(a.astype(np.int16) - b.astype(np.int16)) > 2
    print("%-30s bytes differing f32 vs f64: %.4f %%   (by more than 2 levels: %.4f %%), pixels: %.4f %%"
          % (name, 100 * d.mean(), 100 * big.mean(), 100 * d.any(axis=2).mean()))
