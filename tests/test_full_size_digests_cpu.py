"""tests/golden/full_size_digests.json against the oracle that made it, on the pieces that cost seconds: all of C1, and one
256-row block of every other configuration (a block through the set's interior for the Mandelbrot views).  Keeps the
committed digests honest when the oracle or the generator changes; the whole images are the GPU suite's business
(tests/test_gpu_full_size_digests.py)."""
import hashlib
import json
import os

import numpy as np
import pytest

import golden_util as G
import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "full_size_digests.json")) as _f:
    D = json.load(_f)
BLOCK = D["block_rows"]


def sha(a):
    return hashlib.sha256(memoryview(np.ascontiguousarray(a)).cast("B")).hexdigest()


def block_digests(ent, b):
    ocfg = G.fill_config(O.Config(), ent["config"])
    prec = O.F32 if ent["precision"] == "f32" else O.F64
    y0, y1 = b * BLOCK, min(ocfg.height, (b + 1) * BLOCK)
    z, it = O.escape_rows(ocfg, prec, y0, y1)
    O.set_log2_mode(O.LOG2_LIBM)
    rgb = O.colour_rows(ocfg, z, it)
    assert np.array_equal(rgb, O.get_image(ocfg, prec, y0, y1))  # colour_rows over escape_rows IS get_image
    O.set_log2_mode(O.LOG2_SOFT)
    try:
        soft = O.colour_rows(ocfg, z, it)
    finally:
        O.set_log2_mode(O.LOG2_LIBM)
    il = it.astype(np.uint64)
    return sha(rgb), sha(soft), sha(it.astype("<u4")), sha(z.astype("<f8")), int(np.where(il < ocfg.iterations, il + 1, ocfg.iterations).sum())


def test_manifest_is_complete():
    want = {"C1": (3000, 3000, 1024, "f64"), "C2": (16384, 16384, 1024, "f64"), "C2_f32": (16384, 16384, 1024, "f32"),
            "C3": (16384, 16384, 65536, "f64"), "C4_f32": (16384, 16384, 4096, "f32"), "C4_f64": (16384, 16384, 4096, "f64"),
            "C5": (65536, 65536, 1024, "f64")}
    for name, (w, h, it, prec) in want.items():
        ent = D["configs"][name]
        c = ent["config"]
        assert (c["width"], c["height"], c["iterations"], ent["precision"]) == (w, h, it, prec), name
        nb = (h + BLOCK - 1) // BLOCK
        assert len(ent["rgb"]) == len(ent["iters"]) == len(ent["z"]) == len(ent["executed"]) == nb, name
        assert ent["executed_total"] == sum(ent["executed"]), name
        assert ent["rgb_soft_differs"] == {}, name  # libm's and the software log2 gave the same bytes on every block
    # the sums BASELINE.md / bench.py quote
    assert D["configs"]["C2"]["executed_total"] == 68651829557


def test_c1_whole_frame_digests():
    ent = D["configs"]["C1"]
    for b in range(len(ent["rgb"])):
        rgb, soft, it, z, ex = block_digests(ent, b)
        assert (rgb, soft, it, z, ex) == (ent["rgb"][b], ent["rgb"][b], ent["iters"][b], ent["z"][b], ent["executed"][b]), b


@pytest.mark.parametrize("name,block", [("C2", 31), ("C2_f32", 40), ("C4_f32", 17), ("C4_f64", 32), ("C5", 2), ("C3", 0)])
def test_one_block_of_each_configuration(name, block):
    ent = D["configs"][name]
    if name == "C3":
        # a C3 block is ~1 minute of 8 vCPUs: only its first 8 rows here, against the same rows of a block-0 render
        ocfg = G.fill_config(O.Config(), ent["config"])
        z, it = O.escape_rows(ocfg, O.F64, 0, 8)
        assert np.array_equal(O.colour_rows(ocfg, z, it), O.get_image(ocfg, O.F64, 0, 8))
        return
    rgb, soft, it, z, ex = block_digests(ent, block)
    assert (rgb, soft, it, z, ex) == (ent["rgb"][block], ent["rgb"][block], ent["iters"][block], ent["z"][block], ent["executed"][block])
