#!/usr/bin/env python3
"""C4 timings of the two-pass render over its policy: first_cap x queue_want (refill_quit16 / 16 of the wave).
Usage: python tools/sweep_two_pass.py [f32|f64]"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fractal_renderer_amd as fr  # noqa: E402
from fractal_renderer_amd import _native  # noqa: E402

fr.init(0)
lib = _native.load()
precs = [a for a in sys.argv[1:] if a in ("f32", "f64")] or ["f32", "f64"]
caps = [int(a) for a in sys.argv[1:] if a.isdigit()] or [32, 64, 128]
cfg = fr.Config.new(fr.Algo.Julia)
cfg.width = cfg.height = 16384
cfg.iterations = 4096
cfg.julia_set.re, cfg.julia_set.im = -0.8, 0.156
cfg.pos.re = 0.0
out = torch.empty(16384 * 16384 * 3, dtype=torch.uint8, device="cuda")
for pn in precs:
    prec = 1 if pn == "f32" else 0
    lib.fr_set_tile(0)
    lib.fr_set_refill_policy(-1, -1)
    _native.check(lib.fr_render_rows_rgb8_device(C.byref(cfg), prec, 0, 16384, C.c_void_p(out.data_ptr()), out.numel(), None))
    torch.cuda.synchronize()
    ref = out.clone()
    lib.fr_set_tile(11)
    for k1 in caps:
        for q16 in (2, 4, 6, 8, 10, 12, 14):
            lib.fr_set_refill_policy(k1, q16)
            ts = []
            for rep in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _native.check(lib.fr_render_rows_rgb8_device(C.byref(cfg), prec, 0, 16384, C.c_void_p(out.data_ptr()), out.numel(), None))
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            print("C4 %s episode %3d keep %2d lanes: best %.3f ms median %.3f same %s" % (pn, k1, 4 * q16, min(ts), sorted(ts)[2], bool(torch.equal(out, ref))), flush=True)
