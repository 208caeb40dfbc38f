#!/usr/bin/env python3
"""Would C4 gain from rendering the image as K row chunks on two streams (the second pass of one chunk beside the first pass
of the next)?  Priced with the library as it is: the device-pointer entry point per chunk, chunks alternating between two
streams forked from / joined to a main stream by events, HIP-event time from fork to join; K = 1 is today's single launch.
Usage (GPU box): python tools/c4_split_probe.py [f32|f64]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fractal_renderer_amd as fr  # noqa: E402
from fractal_renderer_amd import _native  # noqa: E402
import bench  # noqa: E402

fr.init(0)
lib = _native.load()
size = int(os.environ.get("C4_SIZE", "16384"))
for pn in (sys.argv[1:] or ["f32", "f64"]):
    prec = fr.Precision.F32 if pn == "f32" else fr.Precision.F64
    cfg = bench.make_config(fr, "julia", size, 4096)
    out = torch.empty(size * size * 3, dtype=torch.uint8, device="cuda")
    ref = None
    main = torch.cuda.current_stream()
    side = [torch.cuda.Stream(), torch.cuda.Stream()]
    for K in (1, 2, 3, 4, 6, 8):
        rows = [(size * k // K) // 32 * 32 for k in range(K)] + [size]
        times = []
        for rep in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main)
            joins = []
            for k in range(K):
                st = main if K == 1 else side[k % 2]
                if st is not main:
                    st.wait_event(e0)
                y0, y1 = rows[k], rows[k + 1]
                _native.check(lib.fr_render_rows_rgb8_device(C.byref(cfg), int(prec), y0, y1, C.c_void_p(out.data_ptr() + 3 * size * y0),
                                                             3 * size * (y1 - y0), C.c_void_p(st.cuda_stream)))
                if st is not main:
                    ej = torch.cuda.Event()
                    ej.record(st)
                    joins.append(ej)
            for ej in joins:
                main.wait_event(ej)
            e1.record(main)
            torch.cuda.synchronize()
            if rep >= 3:
                times.append(e0.elapsed_time(e1))
        if ref is None:
            ref = out.clone()
        ts = sorted(times)
        print("C4 %s %d^2: %d chunk(s) on %s: best %.3f ms median %.3f  bytes identical to one launch: %s" % (
            pn, size, K, "the caller's stream" if K == 1 else "two streams", ts[0], ts[len(ts) // 2], bool(torch.equal(out, ref))), flush=True)
