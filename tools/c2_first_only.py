#!/usr/bin/env python3
"""C2 (Mandelbrot 16384^2 i=1024) through the strip kernel (tile 8) and through the first pass alone (tile 13: freeze and
finish) at episodes of 64 ... 1000 iterations, interleaved in one process: does freeze-and-finish beat the strip kernel's
checked path on a view with an interior?  (profiles/r03_kernel_choice_c2_c3.txt: no.)  Usage (GPU box): python tools/c2_first_only.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fractal_renderer_amd as fr
from fractal_renderer_amd import _native
fr.init(0)
lib = _native.load()
lib.fr_set_profiling(1)
size = 16384
for prec, pn in ((0, "f64"), (1, "f32")):
    cfg = fr.Config.new(fr.Algo.Mandelbrot)
    cfg.width = cfg.height = size
    cfg.iterations = 1024
    variants = [("strips", dict(tile=8)), ("t13 e64", dict(tile=13)), ("t13 e256", dict(tile=13, refill_minrun=256)),
                ("t13 e512", dict(tile=13, refill_minrun=512)), ("t13 e1000", dict(tile=13, refill_minrun=1000))]
    outs = {}
    times = {n: [] for n, _ in variants}
    for rep in range(5):
        for n, kw in variants:
            d = outs.setdefault(n, torch.empty(size * size * 3, dtype=torch.uint8, device="cuda"))
            o = fr.RenderOpts(**kw)
            _native.check(lib.fr_render_rows_rgb8_device_opts(C.byref(cfg), prec, 0, size, C.c_void_p(d.data_ptr()), d.numel(), None, C.byref(o)))
            ms = C.c_float()
            _native.check(lib.fr_last_kernel_ms(C.byref(ms)))
            if rep >= 1:
                times[n].append(ms.value)
    torch.cuda.synchronize()
    for n, _ in variants:
        ts = sorted(times[n])
        print("C2 %s %-10s best %.3f median %.3f same %s" % (pn, n, ts[0], ts[len(ts) // 2], bool(torch.equal(outs[n], outs["strips"]))), flush=True)
