"""Load tests/golden/cases.json + golden_vectors.npz (made by tests/golden/make_golden.py)."""
import json
import os

import numpy as np

import oracle_lib as O

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

with open(os.path.join(GOLDEN_DIR, "cases.json")) as _f:
    MANIFEST = json.load(_f)
KEYS = sorted(MANIFEST)
_npz = None


def vectors():
    global _npz
    if _npz is None:
        _npz = np.load(os.path.join(GOLDEN_DIR, "golden_vectors.npz"))
    return _npz


def fill_config(cfg, d):
    """Fill any ctypes struct with calc::Config's field names from a manifest dict."""
    fh = float.fromhex
    cfg.algo, cfg.width, cfg.height, cfg.iterations = d["algo"], d["width"], d["height"], d["iterations"]
    cfg.limit, cfg.stable_limit = fh(d["limit"]), fh(d["stable_limit"])
    cfg.pos.re, cfg.pos.im = map(fh, d["pos"])
    cfg.scale.re, cfg.scale.im = map(fh, d["scale"])
    cfg.exposure = fh(d["exposure"])
    cfg.inside, cfg.smooth = d["inside"], d["smooth"]
    for name in ("primary_color", "secondary_color"):
        c = getattr(cfg, name)
        c.r, c.g, c.b = d[name]
    cfg.color_weight = fh(d["color_weight"])
    cfg.julia_set.re, cfg.julia_set.im = map(fh, d["julia_set"])
    return cfg


def oracle_config(key):
    return fill_config(O.Config(), MANIFEST[key]["config"])


def precision_of(key):
    return O.F32 if MANIFEST[key]["precision"] == "f32" else O.F64
