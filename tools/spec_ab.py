#!/usr/bin/env python3
"""Speculative long blocks (fr_kernels.hip: FR_SC_SPEC_BODY) against the checked blocks of four, interleaved in one process
so that clock drift hits both alike: loop_mode -1 (automatic: speculation where the host can prove it) against 5 (the same
without speculation), on the BASELINE views at SPEC_SIZE^2 (default 16384) plus the views the default dispatch sends through
other kernels.  Prints best / median kernel ms (HIP events around the launch) and whether the bytes agree.
Usage (GPU box): python tools/spec_ab.py [c2 c2f32 c1 gui4k filled c4 c4f64 c3]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fractal_renderer_amd as fr  # noqa: E402
from fractal_renderer_amd import _native  # noqa: E402

size = int(os.environ.get("SPEC_SIZE", "16384"))
reps = int(os.environ.get("SPEC_REPS", "9"))  # c3: 3
modes = [int(m) for m in os.environ.get("SPEC_MODES", "-1,5").split(",")]
tile = int(os.environ.get("SPEC_TILE", "0"))


import bench  # noqa: E402  (the views of bench.py: the same kernels are chosen)

CASES = {
    # name: (bench view, edge, iterations, precision)
    "c2": ("default", size, 1024, "f64"),
    "c2f32": ("default", size, 1024, "f32"),
    "c1": ("zoom1e6", 3000, 1024, "f64"),
    "c3": ("zoom1e6", size, 65536, "f64"),
    "c4": ("julia", size, 4096, "f32"),
    "c4f64": ("julia", size, 4096, "f64"),
    "filled": ("filled_julia", size, 1024, "f64"),
    "gui4k": ("default", 0, 1024, "f64"),  # 3840 x 2160
    "basilica0": ("filled_julia", size, 1024, "f64"),  # c = -1 + 0i EXACTLY: the unscaled loop (a zero component)
}


# the 13 views of tools/two_pass_views.py (Julia dusts, filled sets, dendrites, Mandelbrot exteriors and boundaries), 8192^2:
# `python tools/spec_ab.py views13` — does speculation ever COST a view?
VIEWS13 = [
    ("julia dust -0.8+0.156i", "julia", (-0.8, 0.156), (0.0, 0.0), 0.4, 4096),
    ("julia rabbit -0.12+0.74i", "julia", (-0.12, 0.74), (0.0, 0.0), 0.4, 1024),
    ("julia 0.285+0.01i", "julia", (0.285, 0.01), (0.0, 0.0), 0.4, 1024),
    ("julia dendrite i", "julia", (0.0, 1.0), (0.0, 0.0), 0.4, 1024),
    ("julia basilica -1", "julia", (-1.0, 1e-9), (0.0, 0.0), 0.4, 1024),
    ("julia siegel -0.391-0.587i", "julia", (-0.391, -0.587), (0.0, 0.0), 0.4, 2048),
    ("julia dust 0.4+0.4i", "julia", (0.4, 0.4), (0.0, 0.0), 0.4, 256),
    ("mandelbrot default view", "mandelbrot", None, (-0.6, 0.0), 0.4, 1024),
    ("mandelbrot exterior, far out", "mandelbrot", None, (0.0, 0.0), 0.1, 1024),
    ("mandelbrot exterior beside the antenna", "mandelbrot", None, (-1.9, 0.15), 4.0, 4096),
    ("mandelbrot seahorse valley edge", "mandelbrot", None, (-0.745, 0.25), 8.0, 4096),
    ("mandelbrot exterior filaments x200", "mandelbrot", None, (-0.7436, 0.1402), 200.0, 4096),
    ("mandelbrot deep boundary 1e6", "mandelbrot", None, (-0.7436447860, 0.1318252536), 1e6, 4096),
]
for _k, (_n, _a, _js, _pos, _sc, _it) in enumerate(VIEWS13):
    for _p in ("f64", "f32"):
        bench.VIEWS["v13_%d" % _k] = (_a, _pos, _sc, _js)
        CASES["v13_%d_%s" % (_k, _p)] = ("v13_%d" % _k, 8192, _it, _p)


def view(name):
    v, edge, its, prec = CASES[name]
    cfg = bench.make_config(fr, v, edge or 3840, its)
    if not edge:
        cfg.height = 2160
    if name == "basilica0":
        cfg.julia_set.im = 0.0
    return cfg, fr.Precision.F32 if prec == "f32" else fr.Precision.F64


fr.init(0)
lib = _native.load()
lib.fr_set_profiling(1)
lib.fr_set_tile(tile)
names = sys.argv[1:] or ["c2", "c2f32", "c1", "gui4k", "filled", "c4", "c4f64", "c3"]
if names == ["views13"]:
    names = ["v13_%d_%s" % (k, p) for p in ("f64", "f32") for k in range(len(VIEWS13))]
for name in names:
    cfg, prec = view(name)
    nbytes = cfg.width * cfg.height * 3
    outs = {m: torch.empty(nbytes, dtype=torch.uint8, device="cuda") for m in modes}
    times = {m: [] for m in modes}
    names = {}
    for rep in range(3 if name == "c3" else reps):
        for m in modes:
            _native.check(lib.fr_set_loop_mode(m))
            _native.check(lib.fr_render_rows_rgb8_device(C.byref(cfg), int(prec), 0, cfg.height, C.c_void_p(outs[m].data_ptr()), nbytes, None))
            ms = C.c_float()
            _native.check(lib.fr_last_kernel_ms(C.byref(ms)))
            if rep >= (1 if name == "c3" else 2):
                times[m].append(ms.value)
            kn = C.create_string_buffer(256)
            lib.fr_last_kernel_name(kn, 256)
            names[m] = kn.value.decode()
    lib.fr_set_loop_mode(-1)
    torch.cuda.synchronize()
    base = sorted(times[modes[-1]])
    for m in modes:
        ts = sorted(times[m])
        same = bool(torch.equal(outs[m], outs[modes[-1]]))
        label = name if not name.startswith("v13_") else "%s %s" % (name.rsplit("_", 1)[1], VIEWS13[int(name.split("_")[1])][0])
        print("%-44s %dx%d i=%d loop_mode %2d: best %8.3f ms  median %8.3f  (%+5.1f %% vs mode %d)  bytes identical: %s  [%s]" % (
            label, cfg.width, cfg.height, cfg.iterations, m, ts[0], ts[len(ts) // 2], 100.0 * (ts[len(ts) // 2] / base[len(base) // 2] - 1.0),
            modes[-1], same, names[m][:60]), flush=True)
    del outs
    torch.cuda.empty_cache()
