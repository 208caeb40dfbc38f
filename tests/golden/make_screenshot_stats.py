#!/usr/bin/env python3
"""Derive a small fixture from the ONE output artefact the reference holds: screenshots/mandelbrot-1000000x.avif
(README.md:9-11; 1000 x 1000, lossy AV1, rescaled, render parameters unrecorded).  It cannot pin bits, but it does
hold two facts that the oracle otherwise asserts from reading source alone:
  * the emitted channel order of the default primary colour RGB::new(40, 40, 255) through color_multiply's g/b swap
    (calc/src/lib.rs:129-139): exterior pixels are BLUE-dominant, R ~ G, B/R ~ 255/40 — not green;
  * `inside = false` (the -d flag) => interior pixels are BLACK (calc/src/lib.rs:233).
Writes channel STATISTICS (not the image) to tests/golden/reference_screenshot_stats.json.  Needs /root/reference and
PIL with AVIF support (present in the build container); the test reads only the committed JSON."""
import json
import os
import sys

import numpy as np
from PIL import Image

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/screenshots/mandelbrot-1000000x.avif"
a = np.asarray(Image.open(src).convert("RGB")).astype(np.int64)
R, G, B = a[..., 0], a[..., 1], a[..., 2]
mx = a.max(axis=2)
ext = (R >= 16) & (B < 250)  # exterior pixels bright enough for a ratio and not saturated in blue
ratio = B[ext] / np.maximum(R[ext], 1)
stats = {
    "source": "screenshots/mandelbrot-1000000x.avif of the reference (decoded with PIL %s)" % Image.__version__,
    "width": int(a.shape[1]), "height": int(a.shape[0]),
    "black_fraction": float((mx <= 8).mean()),
    "exterior_pixels_measured": int(ext.sum()),
    "b_over_r_median": float(np.median(ratio)), "b_over_r_p10": float(np.percentile(ratio, 10)), "b_over_r_p90": float(np.percentile(ratio, 90)),
    "r_minus_g_median": float(np.median(R[ext] - G[ext])), "abs_r_minus_g_p90": float(np.percentile(np.abs(R[ext] - G[ext]), 90)),
    "fraction_green_above_blue": float((G > B + 8).mean()),
    "fraction_blue_saturated": float((B >= 250).mean()),
}
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_screenshot_stats.json")
json.dump(stats, open(out, "w"), indent=1)
print(json.dumps(stats, indent=1))
