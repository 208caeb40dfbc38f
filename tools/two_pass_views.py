#!/usr/bin/env python3
"""The default dispatch against each fixed kernel choice — strips (8), two passes (11), the first pass alone (13) — on a
set of views: Julia dusts, filled Julia sets, dendrites, and Mandelbrot views (the default frame, exteriors at
several zooms, a deep boundary view).  8192^2, kernel time by HIP events (best of 3 after a warm-up).
Prints, per view and precision, every time, the best fixed choice and how far the default is from it."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fractal_renderer_amd as fr  # noqa: E402
from fractal_renderer_amd import _native  # noqa: E402

fr.init(0)
lib = _native.load()
N = int(os.environ.get("VIEWS_SIZE", "8192"))
out = torch.empty(N * N * 3, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream()
J, M = fr.Algo.Julia, fr.Algo.Mandelbrot
# (name, algo, julia c | None, pos, scale, iterations)
VIEWS = [
    ("julia dust -0.8+0.156i", J, (-0.8, 0.156), (0.0, 0.0), 0.4, 4096),
    ("julia rabbit -0.12+0.74i", J, (-0.12, 0.74), (0.0, 0.0), 0.4, 1024),
    ("julia 0.285+0.01i", J, (0.285, 0.01), (0.0, 0.0), 0.4, 1024),
    ("julia dendrite i", J, (0.0, 1.0), (0.0, 0.0), 0.4, 1024),
    ("julia basilica -1", J, (-1.0, 1e-9), (0.0, 0.0), 0.4, 1024),
    ("julia siegel -0.391-0.587i", J, (-0.391, -0.587), (0.0, 0.0), 0.4, 2048),
    ("julia dust 0.4+0.4i", J, (0.4, 0.4), (0.0, 0.0), 0.4, 256),
    ("mandelbrot default view", M, None, (-0.6, 0.0), 0.4, 1024),
    ("mandelbrot exterior, far out", M, None, (0.0, 0.0), 0.1, 1024),
    ("mandelbrot exterior beside the antenna", M, None, (-1.9, 0.15), 4.0, 4096),
    ("mandelbrot seahorse valley edge", M, None, (-0.745, 0.25), 8.0, 4096),
    ("mandelbrot exterior filaments x200", M, None, (-0.7436, 0.1402), 200.0, 4096),
    ("mandelbrot deep boundary 1e6", M, None, (-0.7436447860, 0.1318252536), 1e6, 4096),
]
worst = 0.0
for pn, prec in (("f32", 1), ("f64", 0)):
    for name, algo, js, pos, scale, it in VIEWS:
        cfg = fr.Config.new(algo)
        cfg.width = cfg.height = N
        cfg.iterations = it
        if js:
            cfg.julia_set.re, cfg.julia_set.im = js
        cfg.pos.re, cfg.pos.im = pos
        cfg.scale.re = cfg.scale.im = scale
        cfg.exposure = 5.0
        st = (C.c_double * 8)()
        _native.check(lib.fr_debug_sample_view(C.byref(cfg), prec, st))
        line = "%s %-40s i=%-5d lanes %.3f cap %.2f mean %6.1f handover %.4f waste/work %.3f" % (
            pn, name, it, st[6], st[3] / (64.0 * st[2]), st[0] / (64.0 * st[2]), st[4] / (64.0 * st[2]), st[5] / max(st[0], 1.0))
        ref = None
        t = {}
        for rep in range(2):  # warm-up: clocks and caches settle before the first measured choice
            o = fr.RenderOpts(tile=8)
            _native.check(lib.fr_render_rows_rgb8_device_opts(C.byref(cfg), prec, 0, N, out.data_ptr(), out.numel(), s.cuda_stream, C.byref(o)))
        torch.cuda.synchronize()
        for tile in (0, 8, 11, 13):
            o = fr.RenderOpts(tile=tile)
            ts = []
            for rep in range(6):
                _native.check(lib.fr_set_profiling(1))
                _native.check(lib.fr_render_rows_rgb8_device_opts(C.byref(cfg), prec, 0, N, out.data_ptr(), out.numel(), s.cuda_stream, C.byref(o)))
                ms = C.c_float(0)
                _native.check(lib.fr_last_kernel_ms(C.byref(ms)))
                ts.append(ms.value)
            torch.cuda.synchronize()
            img = out.clone()
            if ref is None:
                ref = img
            t[tile] = sorted(ts[1:])[1]  # second best of five
            line += "  %s %8.3f%s" % ({0: "default", 8: "strips", 11: "2pass", 13: "1st-only"}[tile], t[tile], "" if torch.equal(img, ref) else " DIFFERENT")
        best = min((8, 11, 13), key=lambda k: t[k])
        gap = t[0] / t[best] - 1.0
        worst = max(worst, gap)
        kn = C.create_string_buffer(256)
        line += "   best %s; default %+.1f%%" % ({8: "strips", 11: "2pass", 13: "1st-only"}[best], 100 * gap)
        print(line, flush=True)
print("largest gap of the default to the best fixed choice: %+.1f%%" % (100 * worst))
