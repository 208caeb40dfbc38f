"""The oracle reproduces the committed golden vectors (regression) in both log2 modes."""
import numpy as np
import pytest

import golden_util as G
import oracle_lib as O


@pytest.mark.parametrize("key", G.KEYS)
def test_oracle_matches_golden(key):
    cfg = G.oracle_config(key)
    prec = G.precision_of(key)
    v = G.vectors()
    for mode in (O.LOG2_LIBM, O.LOG2_SOFT):
        O.set_log2_mode(mode)
        try:
            rgb = O.get_image(cfg, prec)
        finally:
            O.set_log2_mode(O.LOG2_LIBM)
        assert np.array_equal(rgb, v[key + "/rgb"]), (key, mode)
    z, it = O.escape_rows(cfg, prec)
    assert np.array_equal(it, v[key + "/iters"])
    if key + "/z" in v.files:
        assert np.array_equal(z.view(np.uint64), v[key + "/z"].view(np.uint64))  # bit pattern, NaN-safe
    assert O.count_iterations(cfg, prec) == G.MANIFEST[key]["executed_iterations"]


def test_row_ranges_concatenate():
    # get_image rows are independent (src/lib.rs:256-267): any row split reassembles the image
    cfg = G.oracle_config("mandelbrot_default/257x193/f64")
    full = O.get_image(cfg)
    parts = [O.get_image(cfg, y0=a, y1=b) for a, b in ((0, 1), (1, 100), (100, 193))]
    assert np.array_equal(np.concatenate(parts), full)
    assert O.get_image(cfg, y0=5, y1=5).shape == (0, 257, 3)
