#!/usr/bin/env python3
"""End-to-end time of the drop-in call: fr_render_rgb8 into a HOST buffer (what get_image returns,
src/lib.rs:253), i.e. kernel + D2H over PCIe.  Reported in DESIGN.md; never bench.py's `value`."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fractal_renderer_amd as fr  # noqa: E402

fr.init(0)
for (w, h, it) in [(750, 500, 50), (1500, 1000, 50), (1920, 1080, 400), (3840, 2160, 400), (3000, 3000, 1024),
                   (16384, 16384, 1024)]:
    cfg = fr.Config.new()
    cfg.width, cfg.height, cfg.iterations = w, h, it
    cfg.pos.re, cfg.exposure = -0.6, 5.0
    fr.get_image(cfg)  # warm: allocations, code load
    ts = []
    img = None
    for _ in range(5):
        del img  # outside the timed region: returning 805 MB to the OS is not part of the call
        t0 = time.perf_counter()
        img = fr.get_image(cfg)  # a FRESH buffer every call, as get_image returns (src/lib.rs:266-267)
        ts.append(time.perf_counter() - t0)
    # same call into a caller buffer that is already resident (the GUI re-renders into one Vec)
    buf = fr.get_image(cfg)
    tr = []
    for _ in range(5):
        t0 = time.perf_counter()
        fr.get_image_rows(cfg, 0, h, out=buf)
        tr.append(time.perf_counter() - t0)
    total, _ = fr.count_iterations(cfg)
    print("%5dx%-5d i=%-5d   ... into a resident buffer: best %.2f ms (%.3e px-it/s)" % (w, h, it, min(tr) * 1e3, total / min(tr)))
    best = min(ts)
    print("%5dx%-5d i=%-5d host-buffer get_image: best %.2f ms, median %.2f ms -> %.3e px-it/s, %.1f Mpx/s, %.2f GB/s out"
          % (w, h, it, best * 1e3, sorted(ts)[2] * 1e3, total / best, w * h / best / 1e6, 3 * w * h / best / 1e9))
