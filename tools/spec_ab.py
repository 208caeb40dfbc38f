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
for name in sys.argv[1:] or ["c2", "c2f32", "c1", "gui4k", "filled", "c4", "c4f64", "c3"]:
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
        print("%-11s %dx%d i=%d loop_mode %2d: best %8.3f ms  median %8.3f  (%+5.1f %% vs mode %d)  bytes identical: %s  [%s]" % (
            name, cfg.width, cfg.height, cfg.iterations, m, ts[0], ts[len(ts) // 2], 100.0 * (ts[len(ts) // 2] / base[len(base) // 2] - 1.0),
            modes[-1], same, names[m][:60]), flush=True)
    del outs
    torch.cuda.empty_cache()
