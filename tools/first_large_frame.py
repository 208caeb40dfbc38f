#!/usr/bin/env python3
"""The drop-in pattern of the reference's GUI (src/gui.rs:56-82): every redraw gets a FRESH buffer from get_image and drops
it after the upload.  Times 100 such calls per frame size through the host-buffer entry point (kernel + D2H + call), the
buffer touched before the call, and 100 calls into one buffer that stays.  FR_TRACE=1 adds the library's own timeline.
Usage (GPU box): python tools/first_large_frame.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import fractal_renderer_amd as fr  # noqa: E402

fr.init(0)
print("FR_HOST_STAGING=%s FR_COPY_THREADS=%s" % (os.environ.get("FR_HOST_STAGING", "(default: on)"), os.environ.get("FR_COPY_THREADS", "(default)")))
for w, h in ((750, 500), (1500, 1000), (1920, 1080), (3840, 2160), (5000, 3000)):
    cfg = fr.Config.new()
    cfg.width, cfg.height, cfg.iterations = w, h, 1024 if w > 1500 else 50
    cfg.pos.re, cfg.exposure = -0.6, 5.0
    keep = np.ones((h, w, 3), dtype=np.uint8)
    t0 = time.perf_counter()
    fr.get_image_rows(cfg, 0, h, fr.Precision.F64, out=keep)
    first = (time.perf_counter() - t0) * 1e3
    same, fresh, lib_fresh = [], [], []
    for _ in range(100):
        t0 = time.perf_counter()
        fr.get_image_rows(cfg, 0, h, fr.Precision.F64, out=keep)
        same.append((time.perf_counter() - t0) * 1e3)
    for _ in range(100):
        buf = np.empty((h, w, 3), dtype=np.uint8)
        buf.fill(1)
        t0 = time.perf_counter()
        fr.get_image_rows(cfg, 0, h, fr.Precision.F64, out=buf)
        fresh.append((time.perf_counter() - t0) * 1e3)
        del buf
    for _ in range(100):
        t0 = time.perf_counter()
        img = fr.get_image(cfg)  # the library's own fresh buffer (untouched pages: first touch inside the call)
        lib_fresh.append((time.perf_counter() - t0) * 1e3)
        del img
    f = lambda v: "min %.2f med %.2f max %.2f" % (min(v), sorted(v)[len(v) // 2], max(v))
    print("%dx%d (%.1f MB): first call %.2f | same buffer %s | fresh touched buffer per call %s | get_image() %s" % (w, h, 3e-6 * w * h, first, f(same), f(fresh), f(lib_fresh)), flush=True)
