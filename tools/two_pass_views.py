#!/usr/bin/env python3
"""The default dispatch (two passes) against the strip (8) and patch-refill (9) kernels on other Julia sets than
C4's dust: filled sets with large interiors, dendrites, high caps.  8192^2, kernel time by HIP events."""
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
N = 8192
out = torch.empty(N * N * 3, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream()
VIEWS = [("dust -0.8+0.156i", (-0.8, 0.156), 4096), ("rabbit -0.12+0.74i", (-0.12, 0.74), 1024), ("0.285+0.01i", (0.285, 0.01), 1024),
         ("dendrite i", (0.0, 1.0), 1024), ("basilica -1", (-1.0, 1e-9), 1024), ("siegel -0.391-0.587i", (-0.391, -0.587), 2048),
         ("dust 0.4+0.4i", (0.4, 0.4), 256)]
for pn, prec in (("f32", 1), ("f64", 0)):
    for name, js, it in VIEWS:
        cfg = fr.Config.new(fr.Algo.Julia)
        cfg.width = cfg.height = N
        cfg.iterations = it
        cfg.julia_set.re, cfg.julia_set.im = js
        cfg.pos.re = 0.0
        line = "%s %-22s i=%-5d" % (pn, name, it)
        ref = None
        for tile in (0, 8, 9):
            o = fr.RenderOpts(tile=tile)
            ts = []
            for rep in range(4):
                _native.check(lib.fr_set_profiling(1))
                _native.check(lib.fr_render_rows_rgb8_device_opts(C.byref(cfg), prec, 0, N, out.data_ptr(), out.numel(), s.cuda_stream, C.byref(o)))
                ms = C.c_float(0)
                _native.check(lib.fr_last_kernel_ms(C.byref(ms)))
                ts.append(ms.value)
            torch.cuda.synchronize()
            img = out.clone()
            if ref is None:
                ref = img
            line += "  tile %d %8.3f ms%s" % (tile, min(ts[1:]), "" if torch.equal(img, ref) else " DIFFERENT")
        print(line, flush=True)
