#!/usr/bin/env python3
"""bench.py's gui_latency frame sequence on its own (two views x four frame shapes x RGB / RGBA, a NEW touched buffer per
shape, first call + ten more), six rounds; calls over 5 ms are flagged.  FR_TRACE=1 adds the library's timeline of the
large frames.  (profiles/r03_gui_fresh_buffer_pattern.txt)  Usage (GPU box): python tools/gui_sequence_trace.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fractal_renderer_amd as fr
fr.init(0)
frames = [(750, 500, 50), (1500, 1000, 50), (1920, 1080, 1024), (3840, 2160, 1024)]
for rnd in range(6):
    for view in ("default", "julia"):
        for w, h, it in frames:
            cfg = fr.Config.new(fr.Algo.Julia if view == "julia" else fr.Algo.Mandelbrot)
            cfg.width, cfg.height, cfg.iterations = w, h, it
            if view == "julia":
                cfg.julia_set.re, cfg.julia_set.im = -0.8, 0.156
            for ch in (3, 4):
                buf = np.zeros((h, w, ch), dtype=np.uint8)
                buf.fill(1)
                sys.stderr.write("== round %d %s %dx%d ch %d\n" % (rnd, view, w, h, ch)); sys.stderr.flush()
                t0 = time.perf_counter()
                if ch == 3:
                    fr.get_image_rows(cfg, 0, h, fr.Precision.F64, out=buf)
                else:
                    fr.get_image_rgba(cfg, fr.Precision.F64, out=buf)
                first = (time.perf_counter() - t0) * 1e3
                ts = []
                for _ in range(10):
                    t0 = time.perf_counter()
                    if ch == 3:
                        fr.get_image_rows(cfg, 0, h, fr.Precision.F64, out=buf)
                    else:
                        fr.get_image_rgba(cfg, fr.Precision.F64, out=buf)
                    ts.append((time.perf_counter() - t0) * 1e3)
                flag = "  <<<<<< STALL" if first > 5 or max(ts) > 5 else ""
                sys.stderr.write("   first %.3f ms, then max %.3f%s\n" % (first, max(ts), flag)); sys.stderr.flush()
