#!/usr/bin/env python3
"""Regenerate tests/golden/cases.json + golden_vectors.npz from the CPU oracle.

The reference (Rust) cannot be built or run in this pipeline and ships no fixtures
(SURVEY.md §4, §8c), so these vectors are produced by oracle/ (libm log2 mode — what the
reference's f64::log2 resolves to on linux-gnu) AFTER its known-answer tests pass.  They are
regression vectors for the oracle and inputs/expected outputs for the HIP parity tests; they
do not pin the oracle to the reference — only the KATs in test_oracle_kat.py do.

Case list follows SURVEY.md §8c "Golden vectors to generate and commit".
Colours are given as the STORED struct fields {r, g, b} of calc::RGB (calc/src/lib.rs:121-131).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

DEEP = dict(scale=(500000.0, 500000.0), pos=(-0.7436447860, 0.1318252536))  # examples.md:29
C1 = dict(scale=(1e6, 1e6), pos=(-0.7436447860, 0.1318252536))  # BASELINE.md C1 view

# name -> (base, overrides).  base "cli" = flags left at CLI defaults (src/lib.rs:34-226),
# base "new" = Config::new (calc/src/lib.rs:39-69).
CASES = {
    "mandelbrot_default": ("cli", dict(iterations=50)),
    "mandelbrot_new_defaults": ("new", dict()),
    "disable_inside": ("cli", dict(iterations=50, inside=0)),
    "unsmooth": ("cli", dict(iterations=50, smooth=0)),
    "golden_fringe_i400": ("cli", dict(iterations=400)),
    "julia_m08_0156": ("cli", dict(algo=O.JULIA, julia_set=(-0.8, 0.156), iterations=200)),
    "julia_0285_001": ("cli", dict(algo=O.JULIA, julia_set=(0.285, 0.01), iterations=100, exposure=10.0)),
    "deep_5e5": ("cli", dict(iterations=4000, inside=0, exposure=5.0, **DEEP)),
    "c1_view_1e6": ("cli", dict(iterations=1024, **C1)),
    "scale_xy_differ": ("cli", dict(iterations=80, scale=(0.3, 0.55))),
    "limit_2": ("cli", dict(iterations=60, limit=2.0)),
    "limit_half": ("cli", dict(iterations=40, limit=0.5, stable_limit=0.1)),
    "stable_limit_half": ("cli", dict(iterations=50, stable_limit=0.5)),
    "stable_limit_half_limit_1": ("cli", dict(iterations=50, stable_limit=0.5, limit=1.0)),
    "exposure_50": ("cli", dict(iterations=50, exposure=50.0)),
    "iterations_1": ("cli", dict(iterations=1)),
    "iterations_0": ("cli", dict(iterations=0)),
    "iterations_3_unsmooth": ("cli", dict(iterations=3, smooth=0)),
    "hex_colours": ("cli", dict(iterations=50, primary_color=(0xFF, 0x00, 0x80), secondary_color=(0x10, 0xC0, 0x20))),
    "huge_limit_nan_orbits": ("cli", dict(iterations=40, limit=1e200)),
    "barnsley_fern_is_black": ("new", dict(algo=O.BARNSLEY_FERN, iterations=10)),
}
F32_CASES = ["mandelbrot_default", "julia_m08_0156", "unsmooth", "limit_2"]
SIZES = [(64, 64), (257, 193)]


def make_config(name, w, h):
    base, kw = CASES[name]
    kw = dict(kw)
    algo = kw.pop("algo", O.MANDELBROT)
    if base == "cli":
        return O.cli_config(w, h, algo, **kw)
    cfg = O.config_new(algo, **kw)
    cfg.width, cfg.height = w, h
    return cfg


def config_to_dict(cfg):
    return dict(
        algo=cfg.algo, width=cfg.width, height=cfg.height, iterations=cfg.iterations,
        limit=cfg.limit.hex(), stable_limit=cfg.stable_limit.hex(),
        pos=[cfg.pos.re.hex(), cfg.pos.im.hex()], scale=[cfg.scale.re.hex(), cfg.scale.im.hex()],
        exposure=cfg.exposure.hex(), inside=cfg.inside, smooth=cfg.smooth,
        primary_color=list(cfg.primary_color.bytes()), secondary_color=list(cfg.secondary_color.bytes()),
        color_weight=cfg.color_weight.hex(), julia_set=[cfg.julia_set.re.hex(), cfg.julia_set.im.hex()],
    )


def main():
    O.set_log2_mode(O.LOG2_LIBM)
    manifest, arrays = {}, {}
    for name in CASES:
        for (w, h) in SIZES:
            for prec, ptag in ((O.F64, "f64"), (O.F32, "f32")):
                if prec == O.F32 and name not in F32_CASES:
                    continue
                key = "%s/%dx%d/%s" % (name, w, h, ptag)
                cfg = make_config(name, w, h)
                rgb = O.get_image(cfg, prec)
                z, it = O.escape_rows(cfg, prec)
                O.set_log2_mode(O.LOG2_SOFT)
                rgb_soft = O.get_image(cfg, prec)
                O.set_log2_mode(O.LOG2_LIBM)
                manifest[key] = dict(config=config_to_dict(cfg), precision=ptag,
                                     executed_iterations=O.count_iterations(cfg, prec),
                                     soft_log2_differing_bytes=int((rgb != rgb_soft).sum()))
                arrays[key + "/rgb"] = rgb
                arrays[key + "/iters"] = it
                if (w, h) == SIZES[0]:
                    arrays[key + "/z"] = z
    with open(os.path.join(HERE, "cases.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "golden_vectors.npz"), **arrays)
    nd = sum(m["soft_log2_differing_bytes"] for m in manifest.values())
    print("cases:", len(manifest), "bytes differing between libm and soft log2:", nd)


if __name__ == "__main__":
    main()
