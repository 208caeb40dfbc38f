#!/usr/bin/env python3
"""BASELINE C2 and C3 (16384^2, f64) under each fixed kernel choice and the default dispatch (kernel time, HIP events)."""
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
N = 16384
out = torch.empty(N * N * 3, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream()
for name, it, pos, scale, reps in (("C2", 1024, (-0.6, 0.0), 0.4, 5), ("C3", 65536, (-0.7436447860, 0.1318252536), 1e6, 2)):
    cfg = fr.Config.new(fr.Algo.Mandelbrot)
    cfg.width = cfg.height = N
    cfg.iterations = it
    cfg.pos.re, cfg.pos.im = pos
    cfg.scale.re = cfg.scale.im = scale
    cfg.exposure = 5.0
    st = (C.c_double * 8)()
    _native.check(lib.fr_debug_sample_view(C.byref(cfg), 0, st))
    print("%s sample: lanes %.3f capped %.3f mean %.1f waste/work %.4f" % (name, st[6], st[3] / (64 * st[2]), st[0] / (64 * st[2]), st[5] / st[0]), flush=True)
    ref = None
    for tile in (8, 0, 11, 13, 8):
        o = fr.RenderOpts(tile=tile)
        ts = []
        for rep in range(reps):
            _native.check(lib.fr_set_profiling(1))
            _native.check(lib.fr_render_rows_rgb8_device_opts(C.byref(cfg), 0, 0, N, out.data_ptr(), out.numel(), s.cuda_stream, C.byref(o)))
            ms = C.c_float(0)
            _native.check(lib.fr_last_kernel_ms(C.byref(ms)))
            ts.append(ms.value)
        kn = C.create_string_buffer(256)
        lib.fr_last_kernel_name(kn, 256)
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        print("%s tile %2d: best %.3f ms  same %s  [%s]" % (name, tile, min(ts), bool(torch.equal(out, ref)), kn.value.decode()[:50]), flush=True)
