#!/usr/bin/env python3
"""Where does the two-pass render start to pay?  Julia (C4's view and cap) at several launch sizes: the default
dispatch, patch refill (9) and two passes (11), kernel time by HIP events.  Runs on the GPU box."""
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
out = torch.empty(16384 * 16384 * 3, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream()
for pn, prec in (("f32", 1), ("f64", 0)):
    for w, h in ((1024, 1024), (2048, 1024), (2048, 2048), (4096, 2048), (16384, 512), (4096, 4096), (16384, 2048), (8192, 8192)):
        cfg = fr.Config.new(fr.Algo.Julia)
        cfg.width, cfg.height, cfg.iterations = w, h, 4096
        cfg.julia_set.re, cfg.julia_set.im = -0.8, 0.156
        cfg.pos.re = 0.0
        line = "%s %5dx%-5d (%7d tiles):" % (pn, w, h, (w // 8) * (h // 8))
        ref = None
        for tile in (0, 8, 9, 11):
            o = fr.RenderOpts(tile=tile)
            ts = []
            for rep in range(6):
                _native.check(lib.fr_set_profiling(1))
                _native.check(lib.fr_render_rows_rgb8_device_opts(C.byref(cfg), prec, 0, h, out.data_ptr(), 3 * w * h, s.cuda_stream, C.byref(o)))
                ms = C.c_float(0)
                _native.check(lib.fr_last_kernel_ms(C.byref(ms)))
                ts.append(ms.value)
            torch.cuda.synchronize()
            img = out[:3 * w * h].clone()
            if ref is None:
                ref = img
            line += "  tile %2d %.3f ms%s" % (tile, min(ts[1:]), "" if torch.equal(img, ref) else " DIFFERENT")
        print(line, flush=True)
