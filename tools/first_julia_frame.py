#!/usr/bin/env python3
"""A fresh process's FIRST large Julia frame (3840x2160, i=1024: two passes by the default dispatch) through the
device-pointer entry point: wall time of the call that enqueues it and of call + synchronise, then the same again —
does the first one wait for the survivor ring's allocation?  Usage (GPU box): python tools/first_julia_frame.py"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fractal_renderer_amd as fr  # noqa: E402
from fractal_renderer_amd import _native  # noqa: E402

fr.init(0)
lib = _native.load()
w, h = 3840, 2160
cfg = fr.Config.new(fr.Algo.Julia)
cfg.width, cfg.height, cfg.iterations = w, h, 1024
cfg.julia_set.re, cfg.julia_set.im = -0.8, 0.156
out = torch.empty(w * h * 3, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for k in range(4):
    t0 = time.perf_counter()
    _native.check(lib.fr_render_rows_rgb8_device(C.byref(cfg), 0, 0, h, C.c_void_p(out.data_ptr()), out.numel(), None))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    name = C.create_string_buffer(160)
    lib.fr_last_kernel_name(name, 160)
    print("call %d: enqueue %.3f ms, done after %.3f ms  [%s]" % (k, (t1 - t0) * 1e3, (t2 - t0) * 1e3, name.value.decode()[:60]), flush=True)
