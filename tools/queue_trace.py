#!/usr/bin/env python3
"""Per-wave trace of the work-queue kernel on BASELINE C4: how long each persistent wave lived, how many
patches / episodes / colour passes / iterations it ran.  Usage: python tools/queue_trace.py [f32|f64] [tile: 10 = over the image, 11 = second pass of the two-pass render] [first_cap]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fractal_renderer_amd as fr  # noqa: E402
from fractal_renderer_amd import _native  # noqa: E402

prec = 1 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else 0
lib = _native.load()
fr.init(0)
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 10
_native.check(lib.fr_set_tile(tile))  # 10: the work-queue kernel over the image; 11: over the first pass's survivor lists
if len(sys.argv) > 3:
    _native.check(lib.fr_set_refill_policy(int(sys.argv[3]), -1))
cfg = fr.Config.new(fr.Algo.Julia)
cfg.width = cfg.height = 16384
cfg.iterations = 4096
cfg.exposure = 5.0
cfg.julia_set.re, cfg.julia_set.im = -0.8, 0.156
need = 3 * cfg.width * cfg.height
img = torch.empty(need, dtype=torch.uint8, device="cuda:0")
trace = torch.zeros(8192 * 16, dtype=torch.int64, device="cuda:0")
s = torch.cuda.current_stream()
for rep in range(3):
    trace.zero_()
    _native.check(lib.fr_debug_set_queue_trace(trace.data_ptr()))
    _native.check(lib.fr_set_profiling(1))
    _native.check(lib.fr_render_rows_rgb8_device(C.byref(cfg), prec, 0, cfg.height, img.data_ptr(), need, s.cuda_stream))
    ms = C.c_float(0)
    _native.check(lib.fr_last_kernel_ms(C.byref(ms)))
    torch.cuda.synchronize()
_native.check(lib.fr_debug_set_queue_trace(None))
t = trace.cpu().numpy().reshape(-1, 16)
t = t[t[:, 1] > 0]
start, end = t[:, 0], t[:, 1]
t0 = start.min()
life = (end - start) / 100.0  # us
print("kernel %.3f ms; %d waves; span %.1f us" % (ms.value, len(t), (end.max() - t0) / 100.0))
print("start offset us: min %.1f med %.1f p90 %.1f max %.1f" % tuple(np.percentile((start - t0) / 100.0, [0, 50, 90, 100])))
print("end   offset us: min %.1f p10 %.1f med %.1f p90 %.1f max %.1f" % tuple(np.percentile((end - t0) / 100.0, [0, 10, 50, 90, 100])))
print("lifetime us:     min %.1f p10 %.1f med %.1f p90 %.1f max %.1f" % tuple(np.percentile(life, [0, 10, 50, 90, 100])))
for name, col in (("patches", 2), ("episodes", 3), ("colour passes", 4), ("iterations", 5)):
    v = t[:, col]
    print("%-14s sum %d  per wave: min %d p10 %d med %d p90 %d max %d" % ((name, v.sum()) + tuple(np.percentile(v, [0, 10, 50, 90, 100]).astype(int))))
idle = t[t[:, 2] == 0]
print("waves that never opened a patch: %d" % len(idle))
late = np.argsort(end)[-5:]
print("last 5 waves to end: start %s end %s patches %s iters %s" % (((start[late] - t0) / 100.0).round(0), ((end[late] - t0) / 100.0).round(0), t[late, 2], t[late, 5]))

# phase cycle counters (s_memtime ticks = shader cycles): the retire phase includes the finishing passes it triggers
u = t.view(np.uint64) if t.dtype != np.uint64 else t
t6, t7 = t[:, 6].astype(np.uint64), t[:, 7].astype(np.uint64)
ph = {"open patch": t6 & np.uint64(0xFFFFFFFF), "refill": t6 >> np.uint64(32), "main loop": t7 & np.uint64(0xFFFFFFFF),
      "retire (incl. finishing)": t7 >> np.uint64(32), "finishing + colour": t[:, 8].astype(np.uint64)}
act = t[:, 2] > 0
tot = sum(v[act].astype(np.float64).sum() for k, v in ph.items() if k != "finishing + colour")
if tot == 0:  # escape_second_kernel (the survivor lists' own kernel, round 3) records no phase cycles; tile 14 / 10 do
    print("(this kernel records no phase cycles)")
    ph = {}
for k, v in ph.items():
    x = v[act].astype(np.float64)
    print("%-26s %6.1f %% of phase cycles, median per wave %.0f" % (k, 100.0 * x.sum() / tot, np.median(x)))
print("wave lifetime in cycles (median): %.0f" % np.median((life[act]) * 1e-6 * 2.3e9))
