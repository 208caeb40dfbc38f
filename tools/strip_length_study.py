#!/usr/bin/env python3
"""Strip length of the strip kernel (1 / 2 / 4 / 7 tiles per one-wave workgroup) against launch size: kernel ms by HIP events
(second best of five) for a handful of views at sizes from 1280x720 to 8192^2, f32 and f64.  What the by-size rule of the
default dispatch (fr_kernels.hip: launch_precision, case 0) is fitted to."""
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
SIZES = [tuple(int(v) for v in s.split("x")) for s in os.environ.get("SIZES", "1280x720,1920x1080,2048x2048,3840x2160,4096x4096,8192x8192").split(",")]
J, M = fr.Algo.Julia, fr.Algo.Mandelbrot
VIEWS = [
    ("mandelbrot default view", M, None, (-0.6, 0.0), 0.4, 1024),
    ("mandelbrot exterior, far out", M, None, (0.0, 0.0), 0.1, 1024),
    ("mandelbrot seahorse valley edge", M, None, (-0.745, 0.25), 8.0, 4096),
    ("mandelbrot exterior filaments x200", M, None, (-0.7436, 0.1402), 200.0, 4096),
    ("mandelbrot deep boundary 1e6", M, None, (-0.7436447860, 0.1318252536), 1e6, 4096),
    ("julia rabbit -0.12+0.74i", J, (-0.12, 0.74), (0.0, 0.0), 0.4, 1024),
    ("julia dust -0.8+0.156i", J, (-0.8, 0.156), (0.0, 0.0), 0.4, 4096),
    ("julia thin dust 0.4+0.4i", J, (0.4, 0.4), (0.0, 0.0), 0.4, 256),
]
s = torch.cuda.current_stream()
for (W, H) in SIZES:
    out = torch.empty(W * H * 3, dtype=torch.uint8, device="cuda")
    tiles = ((W + 7) // 8) * ((H + 7) // 8)
    print("# ---- %dx%d (%d tiles)" % (W, H, tiles), flush=True)
    for pn, prec in (("f32", 1), ("f64", 0)):
        for name, algo, js, pos, scale, it in VIEWS:
            cfg = fr.Config.new(algo)
            cfg.width, cfg.height, cfg.iterations, cfg.exposure = W, H, it, 5.0
            if js:
                cfg.julia_set.re, cfg.julia_set.im = js
            cfg.pos.re, cfg.pos.im = pos
            cfg.scale.re = cfg.scale.im = scale
            t = {}
            for tile in (8, 1, 2, 4, 8):
                o = fr.RenderOpts(tile=tile)
                ts = []
                for _ in range(6):
                    _native.check(lib.fr_set_profiling(1))
                    _native.check(lib.fr_render_rows_rgb8_device_opts(C.byref(cfg), prec, 0, H, out.data_ptr(), out.numel(), s.cuda_stream, C.byref(o)))
                    ms = C.c_float(0)
                    _native.check(lib.fr_last_kernel_ms(C.byref(ms)))
                    ts.append(ms.value)
                t[tile] = sorted(ts[1:])[1]
            best = min(t, key=lambda k: t[k])
            print("%s %-36s  1: %.4f  2: %.4f  4: %.4f  7: %.4f   best %d  (4 vs best %+.0f%%, 7 vs best %+.0f%%, 2 vs best %+.0f%%, 1 vs best %+.0f%%)" % (
                pn, name, t[1], t[2], t[4], t[8], 7 if best == 8 else best, 100 * (t[4] / t[best] - 1), 100 * (t[8] / t[best] - 1),
                100 * (t[2] / t[best] - 1), 100 * (t[1] / t[best] - 1)), flush=True)
    del out
