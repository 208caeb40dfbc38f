#!/usr/bin/env python3
"""Debug aid: BASELINE C5 through N logical devices + peer gather against the single-device render, on the device:
which rows / columns differ, and does the wrong content equal some other row of the reference?"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import fractal_renderer_amd as fr  # noqa: E402
from fractal_renderer_amd import _native  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
EDGE = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
fr.init(0)
lib = _native.load()
cfg = fr.Config.new(fr.Algo.Mandelbrot)
cfg.width = cfg.height = EDGE
cfg.iterations = 1024
cfg.exposure = 5.0
cfg.pos.re = -0.6
nbytes = 3 * EDGE * EDGE
ref = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream()
_native.check(lib.fr_render_rows_rgb8_device(C.byref(cfg), 0, 0, EDGE, ref.data_ptr(), nbytes, s.cuda_stream))
torch.cuda.synchronize()
fr.init_devices([0] * N)
d = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
_native.check(lib.fr_render_rgb8_multi_device(C.byref(cfg), 0, 0, 0, C.c_void_p(d.data_ptr()), nbytes))
torch.cuda.synchronize()
print("stats", fr.multi_stats())
row_bytes = 3 * EDGE
R = ref.view(EDGE, row_bytes)
D = d.view(EDGE, row_bytes)
bad_rows = []
for y0 in range(0, EDGE, 4096):
    ne = (R[y0:y0 + 4096] != D[y0:y0 + 4096]).any(dim=1)
    bad_rows += (torch.nonzero(ne).flatten() + y0).tolist()
print("rows that differ:", len(bad_rows))
if bad_rows:
    runs, a, prev = [], bad_rows[0], bad_rows[0]
    for y in bad_rows[1:]:
        if y != prev + 1:
            runs.append((a, prev))
            a = y
        prev = y
    runs.append((a, prev))
    print("runs of differing rows (first 40):", runs[:40])
    for (ya, yb) in runs[:6]:
        y = ya
        ne = torch.nonzero(R[y] != D[y]).flatten()
        print("row %d (block %d, device %d): %d bytes differ, columns(px) %d..%d; zero bytes in row: %d" % (
            y, y // 256, (y // 256) % N, ne.numel(), int(ne[0]) // 3, int(ne[-1]) // 3, int((D[y] == 0).sum())))
        # does the row equal another row of the reference?
        lo, hi = max(0, y - 4096), min(EDGE, y + 4096)
        eq = (R[lo:hi] == D[y].unsqueeze(0)).all(dim=1)
        print("   equals reference rows:", (torch.nonzero(eq).flatten() + lo).tolist()[:8])
