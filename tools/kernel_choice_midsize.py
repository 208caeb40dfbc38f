#!/usr/bin/env python3
"""GUI-sized launches (what src/gui.rs:56-82 asks for): the default dispatch FROM THE SECOND FRAME ON (the first frame of a
view goes by algorithm and size and posts a non-blocking sample behind its render; the second reads it) against every
fixed kernel — strips of 1 / 2 / 4 / 7 tiles, two passes and the first pass alone with 7- and 4-tile strips — on 13 views x 2
precisions at each size of SIZES (default 1920x1080, 2048x2048, 3840x2160, 4096x4096).  Kernel time by HIP events, second best of
five after a warm-up.  Prints the sample's statistics, every time, what the default launched on its first and on its
second frame, the best fixed choice and how far the default's second frame is from it; all outputs are compared."""
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
SIZES = [tuple(int(v) for v in s.split("x")) for s in os.environ.get("SIZES", "1920x1080,2048x2048,3840x2160,4096x4096").split(",")]
J, M = fr.Algo.Julia, fr.Algo.Mandelbrot
# (name, algo, julia c | None, pos, scale, iterations)
VIEWS = [
    ("julia dust -0.8+0.156i", J, (-0.8, 0.156), (0.0, 0.0), 0.4, 4096),
    ("julia rabbit -0.12+0.74i", J, (-0.12, 0.74), (0.0, 0.0), 0.4, 1024),
    ("julia 0.285+0.01i", J, (0.285, 0.01), (0.0, 0.0), 0.4, 1024),
    ("julia dendrite i", J, (0.0, 1.0), (0.0, 0.0), 0.4, 1024),
    ("julia basilica -1 (filled)", J, (-1.0, 1e-9), (0.0, 0.0), 0.4, 1024),
    ("julia siegel -0.391-0.587i", J, (-0.391, -0.587), (0.0, 0.0), 0.4, 2048),
    ("julia thin dust 0.4+0.4i", J, (0.4, 0.4), (0.0, 0.0), 0.4, 256),
    ("mandelbrot default view", M, None, (-0.6, 0.0), 0.4, 1024),
    ("mandelbrot exterior, far out", M, None, (0.0, 0.0), 0.1, 1024),
    ("mandelbrot exterior beside the antenna", M, None, (-1.9, 0.15), 4.0, 4096),
    ("mandelbrot seahorse valley edge", M, None, (-0.745, 0.25), 8.0, 4096),
    ("mandelbrot exterior filaments x200", M, None, (-0.7436, 0.1402), 200.0, 4096),
    ("mandelbrot deep boundary 1e6", M, None, (-0.7436447860, 0.1318252536), 1e6, 4096),
]
TILES = [1, 2, 4, 8, 11, 15, 13, 16]
NAMES = {0: "default", 1: "strips1", 2: "strips2", 4: "strips4", 8: "strips7", 11: "2pass7", 15: "2pass4", 13: "1st7", 16: "1st4"}
s = torch.cuda.current_stream()


def kernel_name():
    kn = C.create_string_buffer(256)
    _native.check(lib.fr_last_kernel_name(kn, len(kn)))
    n = kn.value.decode()
    for a, b in (("escape_", ""), ("_kernel", ""), ("<double, ", "<"), ("<float, ", "<"), (" strips in episodes, every tile finished in place", ""),
                 (" strips, then persistent waves over the survivor lists", ""), (" tiles", ""), (" tile", "")):
        n = n.replace(a, b)
    return n


for (W, H) in SIZES:
    out = torch.empty(W * H * 3, dtype=torch.uint8, device="cuda")
    worst = 0.0
    print("# ---- %dx%d (%d tiles)" % (W, H, ((W + 7) // 8) * ((H + 7) // 8)), flush=True)
    for pn, prec in (("f32", 1), ("f64", 0)):
        for name, algo, js, pos, scale, it in VIEWS:
            cfg = fr.Config.new(algo)
            cfg.width, cfg.height = W, H
            cfg.iterations = it
            if js:
                cfg.julia_set.re, cfg.julia_set.im = js
            cfg.pos.re, cfg.pos.im = pos
            cfg.scale.re = cfg.scale.im = scale
            cfg.exposure = 5.0
            st = (C.c_double * 8)()
            _native.check(lib.fr_debug_sample_view(C.byref(cfg), prec, st))
            lanes = 64.0 * st[2]
            line = "%s %-40s i=%-5d cap %.3f mean %7.1f handed %.4f waste %.3f lanes %.3f |" % (
                pn, name, it, st[3] / lanes, st[0] / lanes, st[4] / lanes, st[5] / max(st[0], 1.0), st[6])

            def render(tile):
                o = fr.RenderOpts(tile=tile)
                _native.check(lib.fr_render_rows_rgb8_device_opts(C.byref(cfg), prec, 0, H, out.data_ptr(), out.numel(), s.cuda_stream, C.byref(o)))

            def once(tile):
                _native.check(lib.fr_set_profiling(1))
                render(tile)
                ms = C.c_float(0)
                _native.check(lib.fr_last_kernel_ms(C.byref(ms)))
                return ms.value

            for _ in range(2):  # clocks and caches settle
                render(8)
            torch.cuda.synchronize()
            _native.check(lib.fr_set_profiling(1))
            render(0)  # the view's FIRST frame: by size; the sample is posted behind it
            first_kernel = kernel_name()
            torch.cuda.synchronize()
            time.sleep(0.003)
            once(0)
            second_kernel = kernel_name()
            ref = out.clone()
            # every candidate once per round, six rounds, interleaved: a drifting clock hits all of them alike
            # (measured block after block, the SAME kernel read 4-10 % apart)
            ts = {k: [] for k in [0] + TILES}
            for rnd in range(6):
                for tile in [0] + TILES:
                    ts[tile].append(once(tile))
                    if rnd == 0 and not torch.equal(out, ref):
                        line += " %s DIFFERENT" % NAMES[tile]
            torch.cuda.synchronize()
            t = {k: sorted(v[1:])[1] for k, v in ts.items()}  # second best of the last five
            best = min(TILES, key=lambda k: t[k])
            gap = t[0] / t[best] - 1.0
            worst = max(worst, gap)
            line += " " + " ".join("%s %.4f" % (NAMES[k], t[k]) for k in [0] + TILES)
            line += " | frame1 %s, frame2+ %s; best %s; default %+.1f%%" % (first_kernel, second_kernel, NAMES[best], 100 * gap)
            print(line, flush=True)
    print("largest gap of the default (second frame on) to the best fixed choice at %dx%d: %+.1f%%" % (W, H, 100 * worst), flush=True)
    del out
