#!/usr/bin/env python3
"""C4 (Julia 16384^2 i=4096) kernel time of kernel variants, interleaved so that clock drift hits them alike.
Usage (GPU box): python tools/c4_ab.py [tiles...]   default: 11 14 12   (11 = two passes, 14 = with round 2's second pass,
12 = with round 2's two kernels, 13 = the first pass alone)
Prints per precision: best / median ms per variant (HIP events around the launch) and whether bytes agree."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fractal_renderer_amd as fr  # noqa: E402
from fractal_renderer_amd import _native  # noqa: E402

tiles = [int(a) for a in sys.argv[1:]] or [11, 14, 12]
fr.init(0)
lib = _native.load()
lib.fr_set_profiling(1)
size = int(os.environ.get("C4_SIZE", "16384"))
for prec, pn in ((fr.Precision.F32, "f32"), (fr.Precision.F64, "f64")):
    cfg = fr.Config.new(fr.Algo.Julia)
    cfg.width = cfg.height = size
    cfg.iterations = 4096
    cfg.julia_set.re, cfg.julia_set.im = -0.8, 0.156
    cfg.pos.re = 0.0
    cfg.exposure = 5.0
    outs = {t: torch.empty(size * size * 3, dtype=torch.uint8, device="cuda") for t in tiles}
    times = {t: [] for t in tiles}
    names = {}
    for rep in range(9):
        for t in tiles:
            lib.fr_set_tile(t)
            _native.check(lib.fr_render_rows_rgb8_device(C.byref(cfg), int(prec), 0, size, C.c_void_p(outs[t].data_ptr()), outs[t].numel(), None))
            ms = C.c_float()
            _native.check(lib.fr_last_kernel_ms(C.byref(ms)))
            if rep >= 3:  # the first three warm the survivor-list ring
                times[t].append(ms.value)
            kn = C.create_string_buffer(256)
            lib.fr_last_kernel_name(kn, 256)
            names[t] = kn.value.decode()
    lib.fr_set_tile(0)
    torch.cuda.synchronize()
    for t in tiles:
        ts = sorted(times[t])
        same = bool(torch.equal(outs[t], outs[tiles[0]]))
        print("C4 %s %d^2 tile %2d: best %.3f ms  median %.3f  identical to tile %d: %s  [%s]" % (
            pn, size, t, ts[0], ts[len(ts) // 2], tiles[0], same, names[t][:70]), flush=True)
