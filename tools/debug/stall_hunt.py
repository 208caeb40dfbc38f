#!/usr/bin/env python3
"""Where do the 6-8 ms first-call stalls of GUI-sized frames in bench.py come from?  Rounds of: [optionally a big host render
into a fresh 805 MB buffer, dropped] then first + second calls of four frame shapes into fresh touched buffers, wall time.
FR_TRACE=1 prints the library's enqueue marks for slow calls.  argv: rounds, big (0/1), torch first (0/1)."""
import os
import sys
import time

ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
BIG = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if len(sys.argv) > 3 and sys.argv[3] == "1":
    import torch  # noqa: F401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import fractal_renderer_amd as fr  # noqa: E402

fr.init(0)
shapes = [(750, 500, 50), (1500, 1000, 50), (1920, 1080, 1024), (3840, 2160, 1024)]
for rnd in range(ROUNDS):
    if BIG:
        cfg = fr.Config.new()
        cfg.width = cfg.height = 16384
        cfg.iterations, cfg.exposure = 1024, 5.0
        cfg.pos.re = -0.6
        for _ in range(2):
            big = np.empty((16384, 16384, 3), dtype=np.uint8)
            t0 = time.perf_counter()
            fr.get_image_rows(cfg, 0, 16384, fr.Precision.F64, out=big)
            tb = (time.perf_counter() - t0) * 1e3
            del big
        print("round %d: big host render %.1f ms" % (rnd, tb), flush=True)
    for w, h, it in shapes:
        for ch in (3, 4):
            cfg = fr.Config.new()
            cfg.width, cfg.height, cfg.iterations, cfg.exposure = w, h, it, 5.0
            cfg.pos.re = -0.6 + 1e-6 * rnd
            buf = np.zeros((h, w, ch), dtype=np.uint8)
            buf.fill(1)
            ts = []
            for _ in range(4):
                t0 = time.perf_counter()
                if ch == 3:
                    fr.get_image_rows(cfg, 0, h, fr.Precision.F64, out=buf)
                else:
                    fr.get_image_rgba(cfg, fr.Precision.F64, out=buf)
                ts.append((time.perf_counter() - t0) * 1e3)
            flag = "  <-- STALL" if max(ts) > 2.5 else ""
            print("round %d %dx%d ch%d: %s%s" % (rnd, w, h, ch, " ".join("%.3f" % t for t in ts), flag), flush=True)
            del buf
