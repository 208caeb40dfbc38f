#!/usr/bin/env python3
"""A GEOMETRIC pin from the only output artefact the reference holds: screenshots/mandelbrot-1000000x.avif
(README.md:9-11: a 3000 x 3000 render "at 1,000,000x zoom", stored lossy at 1000 x 1000, parameters unrecorded).

Searching scale x iterations with the oracle shows that the screenshot IS examples.md:29's view rendered square —
`-s 500000 -x -.7436447860 -y .1318252536 -i 4000 -d` — : the oracle's black (interior, `inside = false`) mask overlaps the
screenshot's with IoU 0.98 in the reference's orientation and 0.51 / 0.03 / 0.21 / 0.01 when flipped in y / flipped in x /
transposed / rotated by 180 degrees.  That pins, against something the REFERENCE produced, what the oracle otherwise asserts from
reading source alone: `x / height` (calc/src/lib.rs:194: both axes divided by the HEIGHT), y growing downward (:195), the
centre convention `(coord / max - offset) / scale + pos` (:182-184) and the meaning of --scale.  Parity stays "unpinned" in the
task's sense (a lossy, rescaled image pins no bits); this is the most the artefact can give.

Writes tests/golden/reference_screenshot_geometry.json: the screenshot's black mask reduced to 125 x 125 cells (a cell is
black when at least half of its 8 x 8 pixels are), the IoU table of the parameter search, and how well the oracle's COLOUR
render (3000 x 3000 box-filtered to 1000 x 1000) matches the screenshot.  Needs /root/reference and PIL with AVIF support
(the build container has both); the test (tests/test_oracle_kat.py) reads only the committed JSON."""
import json
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

POS = (-0.7436447860, 0.1318252536)  # examples.md:29
src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/screenshots/mandelbrot-1000000x.avif"
shot = np.asarray(Image.open(src).convert("RGB")).astype(np.float64)
assert shot.shape == (1000, 1000, 3)
black = shot.max(axis=2) <= 8


def cells(mask, n):
    k = mask.shape[0] // n
    return mask[: n * k, : n * k].reshape(n, k, n, k).mean(axis=(1, 3)) >= 0.5


def iou(a, b):
    return float((a & b).sum() / max((a | b).sum(), 1))


def oracle_mask(scale, it, n=500):
    cfg = O.cli_config(n, n, O.MANDELBROT, iterations=it, scale=(scale, scale), pos=POS, inside=0)
    _, iters = O.escape_rows(cfg)
    return iters >= it  # the pixels `inside = false` paints BLACK (calc/src/lib.rs:233)


ref500 = cells(black, 500)
table = []
for scale in (2.5e5, 4e5, 5e5, 1e6, 2e6):
    for it in (1000, 2000, 4000, 8000):
        m = oracle_mask(scale, it)
        table.append({"scale": scale, "iterations": it, "identity": iou(m, ref500), "flip_y": iou(m[::-1], ref500),
                      "flip_x": iou(m[:, ::-1], ref500), "transpose": iou(m.T, ref500), "rot180": iou(m[::-1, ::-1], ref500)})
        print(table[-1], flush=True)
best = max(table, key=lambda r: r["identity"])

colour = []
for it, e in ((4000, 5.0), (4000, 3.0), (8000, 5.0)):
    cfg = O.cli_config(3000, 3000, O.MANDELBROT, iterations=it, scale=(best["scale"],) * 2, pos=POS, inside=0, exposure=e)
    img = O.get_image(cfg).astype(np.float64).reshape(1000, 3, 1000, 3, 3).mean(axis=(1, 3))
    row = {"iterations": it, "exposure": e}
    for name, v in (("identity", img), ("flip_y", img[::-1]), ("flip_x", img[:, ::-1]), ("transpose", img.transpose(1, 0, 2))):
        mse = ((v - shot) ** 2).mean()
        row[name] = {"mean_abs_diff": float(np.abs(v - shot).mean()), "corr_blue": float(np.corrcoef(v[..., 2].ravel(), shot[..., 2].ravel())[0, 1]),
                     "psnr_db": float(10 * np.log10(255.0 ** 2 / mse))}
    colour.append(row)
    print(row, flush=True)

mask125 = cells(black, 125)
doc = {
    "source": "screenshots/mandelbrot-1000000x.avif of the reference (decoded with PIL %s); README.md:9-11" % Image.__version__,
    "view": {"pos": [float.hex(POS[0]), float.hex(POS[1])], "scale": best["scale"], "iterations": best["iterations"], "inside": 0,
             "note": "examples.md:29 (-s 500000 -x -.7436447860 -y .1318252536 -i 4000 -d) rendered square"},
    "black_fraction_screenshot": float(black.mean()),
    "mask_cells": 125,
    "mask_rows_hex": ["%032x" % int("".join("1" if c else "0" for c in row), 2) for row in mask125],
    "iou_search": table,
    "best": best,
    "colour_fit": colour,
}
out = os.path.join(HERE, "reference_screenshot_geometry.json")
json.dump(doc, open(out, "w"), indent=1)
print("best:", best)
