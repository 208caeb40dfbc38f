#!/usr/bin/env python3
"""Per-256-row-block SHA-256 digests of the BASELINE.json configurations at FULL size, from the CPU oracle.

Why: the whole C3 image is 7e12 pixel-iterations (about an hour of this container's 8 vCPUs) and C5 is 12.9 GB,
so the GPU box cannot afford to re-run the oracle on them; until round 4 they were compared on every 64th pixel
in x and y (1/4096 of the image).  This script runs the ORACLE once, here, over every pixel and commits what a
GPU test needs to check the whole image at zero CPU cost on the GPU box: for every block of 256 rows

    rgb    SHA-256 of the packed r,g,b bytes get_image returns for those rows (src/lib.rs:253-270);
    iters  SHA-256 of the u32 (little-endian) escape indices `recursive` returns, row-major (calc/src/lib.rs:245-257);
    z      SHA-256 of the f64 (little-endian, re,im interleaved) final positions it returns;
    executed  the exact sum of executed loop iterations.

One escape pass per block (fro_escape_rows) is coloured twice (fro_colour_rows), with the software log2 the
kernels carry and with the platform libm's log2 — what the reference's f64::log2 calls (calc/src/lib.rs:222-223);
`rgb` is the libm digest and `rgb_soft_differs` lists the blocks (none so far) whose soft-mode bytes differ, with
their soft digest.  The digests are data (hashes of oracle outputs), not reference source.  The oracle is
"parity unpinned" in the task's sense: these vectors pin the GPU path to the oracle, whole image.

Resumable: the JSON is rewritten after every block; a block already present is skipped.
Usage: python tests/golden/make_full_size_digests.py [names...]   (default: every configuration, C3 last)
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402
from make_golden import config_to_dict  # noqa: E402

OUT = os.path.join(HERE, "full_size_digests.json")
BLOCK_ROWS = 256
ZOOM = dict(scale=(1e6, 1e6), pos=(-0.7436447860, 0.1318252536))  # examples.md:29's centre, BASELINE.md "zoom 1e6"

# name -> (width, height, algo, precision, overrides); BASELINE.json `configs` in order, SURVEY.md §8d's flags
CONFIGS = {
    "C1": (3000, 3000, O.MANDELBROT, "f64", dict(iterations=1024, **ZOOM)),
    "C2": (16384, 16384, O.MANDELBROT, "f64", dict(iterations=1024)),
    "C2_f32": (16384, 16384, O.MANDELBROT, "f32", dict(iterations=1024)),
    "C4_f32": (16384, 16384, O.JULIA, "f32", dict(iterations=4096, julia_set=(-0.8, 0.156))),
    "C4_f64": (16384, 16384, O.JULIA, "f64", dict(iterations=4096, julia_set=(-0.8, 0.156))),
    "C5": (65536, 65536, O.MANDELBROT, "f64", dict(iterations=1024)),
    "C3": (16384, 16384, O.MANDELBROT, "f64", dict(iterations=65536, **ZOOM)),
}


def sha(a):
    return hashlib.sha256(memoryview(np.ascontiguousarray(a)).cast("B")).hexdigest()


def save(doc):
    tmp = OUT + ".tmp"
    with open(tmp, "w") as f:
        json.dump(doc, f, indent=0, sort_keys=True)
    os.replace(tmp, OUT)


def main():
    names = sys.argv[1:] or list(CONFIGS)
    doc = json.load(open(OUT)) if os.path.exists(OUT) else {}
    doc.setdefault("block_rows", BLOCK_ROWS)
    doc.setdefault("configs", {})
    for name in names:
        w, h, algo, ptag, kw = CONFIGS[name]
        cfg = O.cli_config(w, h, algo, **kw)
        prec = O.F32 if ptag == "f32" else O.F64
        nblocks = (h + BLOCK_ROWS - 1) // BLOCK_ROWS
        ent = doc["configs"].setdefault(name, dict(config=config_to_dict(cfg), precision=ptag, rgb=[], iters=[], z=[],
                                                   executed=[], rgb_soft_differs={}, oracle_seconds=0.0))
        assert ent["config"] == config_to_dict(cfg) and ent["precision"] == ptag, name
        for b in range(len(ent["rgb"]), nblocks):
            t0 = time.time()
            y0, y1 = b * BLOCK_ROWS, min(h, (b + 1) * BLOCK_ROWS)
            z, it = O.escape_rows(cfg, prec, y0, y1)
            O.set_log2_mode(O.LOG2_LIBM)
            rgb = O.colour_rows(cfg, z, it)
            O.set_log2_mode(O.LOG2_SOFT)
            soft = O.colour_rows(cfg, z, it)
            O.set_log2_mode(O.LOG2_LIBM)
            if b % 16 == 3:  # fro_colour_rows over fro_escape_rows is get_image: spot-check on one row of the block
                assert np.array_equal(O.get_image(cfg, prec, y0 + 5, y0 + 6), rgb[5:6])
            d = sha(rgb)
            ds = sha(soft)
            if ds != d:
                ent["rgb_soft_differs"][str(b)] = ds
            itl = it.astype(np.uint64)
            executed = int(np.where(itl < cfg.iterations, itl + 1, cfg.iterations).sum())
            ent["rgb"].append(d)
            ent["iters"].append(sha(it.astype("<u4")))
            ent["z"].append(sha(z.astype("<f8")))
            ent["executed"].append(executed)
            ent["oracle_seconds"] = round(ent["oracle_seconds"] + time.time() - t0, 2)
            save(doc)
            print("%s block %d/%d  %.1f s  executed %d" % (name, b + 1, nblocks, time.time() - t0, executed), flush=True)
        ent["executed_total"] = sum(ent["executed"])
        save(doc)
    print("done:", ", ".join("%s %d blocks" % (n, len(doc["configs"][n]["rgb"])) for n in doc["configs"]))


if __name__ == "__main__":
    main()
