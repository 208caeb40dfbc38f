"""Known-answer tests that pin the CPU oracle to the reference's source text.

The reference has no tests or fixtures; every expected value below is derived by
hand from calc/src/lib.rs with exactly representable inputs.
"""
import math

import numpy as np
import pytest

import oracle_lib as O


def test_kat1_escape_index_and_position():
    # recursive (calc/src/lib.rs:245-257): 2 -> 6 -> 38 -> 1446 -> 2090918; 2090918^2 > 2^32 at i = 3
    assert O.recursive(50, (2, 0), (2, 0), 65536) == ((2090918.0, 0.0), 3)


def test_kat2_limit_is_a_radius_squared_inside():
    # limit = 2 => squared = 4; first `next` = 6, 36 > 4 => (next, 0)
    assert O.recursive(50, (2, 0), (2, 0), 2) == ((6.0, 0.0), 0)


def test_kat3_bounded_orbit_returns_previous_and_iterations():
    # c = -2: z = 2 forever, never > 2^32; exhaustion returns (previous, iterations) (:256)
    assert O.recursive(50, (-2, 0), (-2, 0), 65536) == ((2.0, 0.0), 50)


def test_kat3_bounded_orbit_is_coloured_as_outside():
    # dist = 4 > stable_limit = 2 (stable_limit is compared with the SQUARED distance, :216)
    cfg = O.config_new(width=4, height=4, scale=(0.25, 0.25))
    # pixel (0, 2) is exactly -2 + 0i: x/H - (W/H)/2 = 0 - 0.5, /0.25 = -2; y: 2/4 - 0.5 = 0
    z = O.lib().fro_xy_to_imaginary(cfg, 0, 2)
    assert (z.re, z.im) == (-2.0, 0.0)
    # smooth: log_zn = log2(2)/2 = 0.5, nu = -1, it = 52, mult = 52/50*2 = 2.08 -> (83, 83, 255)
    assert O.get_recursive_pixel(cfg, 0, 2) == (83, 83, 255)
    cfg.smooth = 0  # mult = 50/50*2 = 2 -> (80, 80, 255 saturated)
    assert O.get_recursive_pixel(cfg, 0, 2) == (80, 80, 255)


def test_kat4_period_two_orbit():
    assert O.recursive(50, (-1, 0), (-1, 0), 65536) == ((-1.0, 0.0), 50)
    assert O.recursive(51, (-1, 0), (-1, 0), 65536) == ((0.0, 0.0), 51)
    cfg = O.config_new(width=4, height=4, scale=(0.25, 0.25))
    assert O.get_recursive_pixel(cfg, 1, 2) == (240, 170, 0)  # inside colour: secondary * dist(=1)
    cfg.iterations = 51
    assert O.get_recursive_pixel(cfg, 1, 2) == (0, 0, 0)  # secondary * 0


def test_kat5_origin():
    for n in (0, 1, 50):
        assert O.recursive(n, (0, 0), (0, 0), 65536) == ((0.0, 0.0), n)


def test_kat6_rgb_new_argument_order():
    # RGB::new(r, b, g) (calc/src/lib.rs:129-131): the SECOND argument is blue
    c = O.lib().fro_rgb_new(40, 40, 255)
    assert (c.r, c.g, c.b) == (40, 255, 40)
    c = O.lib().fro_rgb_new(240, 170, 0)
    assert (c.r, c.g, c.b) == (240, 0, 170)
    cfg = O.config_new()
    assert cfg.primary_color.bytes() == (40, 255, 40)
    assert cfg.secondary_color.bytes() == (240, 0, 170)


def test_config_new_defaults():
    cfg = O.config_new()
    assert (cfg.width, cfg.height, cfg.iterations) == (2000, 1000, 50)
    assert (cfg.limit, cfg.stable_limit, cfg.exposure) == (65536.0, 2.0, 2.0)
    assert (cfg.scale.re, cfg.scale.im) == (0.4, 0.4)
    assert (cfg.inside, cfg.smooth) == (1, 1)
    fern = O.config_new(O.BARNSLEY_FERN)
    assert fern.iterations == 10_000_000
    assert fern.primary_color.bytes() == (4, 3, 100)  # new(4, 100, 3) -> {r4, g3, b100}


KAT7_SMOOTH = [
    [(0, 0, 5), (1, 1, 8), (1, 1, 9), (0, 0, 6)],
    [(1, 1, 12), (3, 3, 22), (240, 170, 0), (2, 2, 13)],
    [(83, 83, 255), (240, 170, 0), (0, 0, 0), (2, 2, 18)],
    [(1, 1, 12), (3, 3, 22), (240, 170, 0), (2, 2, 13)],
]
KAT7_UNSMOOTH = [
    [(4, 4, 30)] * 4,
    [(6, 6, 40), (8, 8, 51), (240, 170, 0), (6, 6, 40)],
    [(80, 80, 255), (240, 170, 0), (0, 0, 0), (6, 6, 40)],
    [(6, 6, 40), (8, 8, 51), (240, 170, 0), (6, 6, 40)],
]


@pytest.mark.parametrize("mode", [O.LOG2_LIBM, O.LOG2_SOFT])
def test_kat7_4x4_image(mode):
    # W = H = 4, scale 0.25: pixel starts are exactly {-2,-1,0,1}^2; Config::new colours, exposure 2
    O.set_log2_mode(mode)
    try:
        cfg = O.config_new(width=4, height=4, scale=(0.25, 0.25))
        img = O.get_image(cfg)
        assert img.shape == (4, 4, 3)
        assert [[tuple(p) for p in row] for row in img.tolist()] == KAT7_SMOOTH
        cfg.smooth = 0  # libm-free variant
        img = O.get_image(cfg)
        assert [[tuple(p) for p in row] for row in img.tolist()] == KAT7_UNSMOOTH
    finally:
        O.set_log2_mode(O.LOG2_LIBM)


def test_x_is_divided_by_height():
    # xy_to_imaginary (calc/src/lib.rs:186-197): re = (x/H - (W/H)/2)/scale + pos — square pixels
    cfg = O.config_new(width=8, height=4, scale=(0.5, 0.5), pos=(0.25, -0.125))
    z = O.lib().fro_xy_to_imaginary(cfg, 8, 4)  # one past the last pixel: x/H = 2, offset 1
    assert (z.re, z.im) == ((2.0 - 1.0) / 0.5 + 0.25, (1.0 - 0.5) / 0.5 - 0.125)
    z = O.lib().fro_xy_to_imaginary(cfg, 0, 0)
    assert (z.re, z.im) == (-1.0 / 0.5 + 0.25, -0.5 / 0.5 - 0.125)


def test_as_u8_saturation_nan_and_negative():
    # exposure huge -> saturate at 255; stable_limit < dist < 1 -> log2(negative) = NaN -> 0
    cfg = O.config_new(width=4, height=4, scale=(0.25, 0.25), exposure=1e9)
    assert O.get_recursive_pixel(cfg, 0, 2) == (255, 255, 255)
    cfg = O.config_new(width=4, height=4, scale=(0.25, 0.25), stable_limit=0.5, iterations=50)
    # pixel (1,2) is c = -1: final z = -1, dist = 1 > 0.5: log_zn = log2(1)/2 = 0, nu = -inf, it = +inf
    assert O.get_recursive_pixel(cfg, 1, 2) == (255, 255, 255)
    # negative multiplier -> 0 (Rust `as u8` saturates at 0)
    cfg = O.config_new(width=4, height=4, scale=(0.25, 0.25), exposure=-3.0)
    assert O.get_recursive_pixel(cfg, 0, 2) == (0, 0, 0)


def test_julia_uses_constant_c():
    # Julia: c = config.julia_set for every pixel (calc/src/lib.rs:210)
    cfg = O.config_new(O.JULIA, width=4, height=4, scale=(0.25, 0.25), julia_set=(-1.0, 0.0), iterations=51)
    # pixel (2,2) starts at 0: 0 -> -1 -> 0 ... after 51 steps z = -1 (odd count)
    z, it = O.escape_rows(cfg)
    assert it[2, 2] == 51 and tuple(z[2, 2]) == (-1.0, 0.0)
    pos, n = O.recursive(51, (0, 0), (-1, 0), 65536)
    assert (pos, n) == ((-1.0, 0.0), 51)


def test_fern_is_black_on_this_path():
    cfg = O.config_new(O.BARNSLEY_FERN, width=5, height=3)
    assert not O.get_image(cfg).any()


def test_f32_fast_path_definition():
    # build-defined (no f32 in the reference): same op order in binary32
    f = np.float32
    re, im = f(0.3), f(0.5)
    cre, cim = re, im
    for i in range(5):
        nre = f(f(f(re * re) - f(im * im)) + cre)
        nim = f(f(f(f(2.0) * re) * im) + cim)
        re, im = nre, nim
    pos, n = O.recursive(5, (0.3, 0.5), (0.3, 0.5), 65536, f32=True)
    assert n == 5 and pos == (float(re), float(im))


def test_executed_iteration_count_definition():
    # escape at index i executed i+1 iterations; exhaustion executed `iterations`
    cfg = O.config_new(width=4, height=4, scale=(0.25, 0.25))
    _, it = O.escape_rows(cfg)
    expect = int(np.where(it < 50, it.astype(np.int64) + 1, 50).sum())
    assert O.count_iterations(cfg) == expect
    total, npx, _ = O.sample_image(cfg, 1, 1)
    assert (total, npx) == (expect, 16)


def test_soft_log2_tracks_libm():
    rng = np.random.default_rng(7)
    xs = np.concatenate([np.exp(rng.uniform(-700, 700, 5000)), rng.uniform(0.5, 2, 5000),
                         1 + rng.uniform(-0.05, 0.05, 5000), rng.uniform(8, 16, 5000)])
    worst = 0
    for x in xs.tolist():
        O.set_log2_mode(O.LOG2_LIBM)
        a = np.float64(O.log2(x)).view(np.int64)
        O.set_log2_mode(O.LOG2_SOFT)
        b = np.float64(O.log2(x)).view(np.int64)
        worst = max(worst, abs(int(a) - int(b)))
    O.set_log2_mode(O.LOG2_LIBM)
    assert worst <= 1
    for x, want in [(0.0, -math.inf), (-0.0, -math.inf), (math.inf, math.inf), (1.0, 0.0), (2.0, 1.0),
                    (0.5, -1.0), (65536.0, 16.0), (5e-324, -1074.0)]:
        O.set_log2_mode(O.LOG2_SOFT)
        assert O.log2(x) == want
    assert math.isnan(O.log2(-1.0)) and math.isnan(O.log2(math.nan))
    O.set_log2_mode(O.LOG2_LIBM)


def test_oracle_soft_log2_is_a_verbatim_copy():
    """The oracle builds from oracle/ alone (VERDICT r02 weak #1b): its soft-mode log2 is a COPY of the product's
    fr_math.h + table.  The copy must not drift: the table byte for byte, the header from its original first
    line on."""
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prod = os.path.join(root, "fractal-renderer_amd", "csrc")
    with open(os.path.join(root, "oracle", "fr_log2_table.inc"), "rb") as a, open(os.path.join(prod, "fr_log2_table.inc"), "rb") as b:
        assert a.read() == b.read()
    copy = open(os.path.join(root, "oracle", "soft_log2.h")).read()
    orig = open(os.path.join(prod, "fr_math.h")).read()
    marker = " * fr_math.h — arithmetic shared by host and device code of the colour-mapping pass."
    assert marker in copy and marker in orig
    assert copy[copy.index(marker):] == orig[orig.index(marker):]
    mk = open(os.path.join(root, "oracle", "Makefile")).read()
    assert "fractal-renderer_amd" not in mk.replace("# ", ""), "oracle/Makefile must not reach into the product tree"
    src = open(os.path.join(root, "oracle", "fractal_oracle.c")).read()
    assert '#include "../' not in src


def test_reference_screenshot_agrees_with_the_oracles_channel_order_and_inside_rule():
    """VERDICT r02 #7.  The reference holds no test vectors; its one output artefact is a lossy, rescaled AVIF whose
    parameters are unrecorded (README.md:9-11).  tests/golden/make_screenshot_stats.py reduced it to channel statistics.
    This is a CONSISTENCY check on SURVEY §8 row a6 — not a pin; parity stays "unpinned":
      * exterior pixels of the screenshot are blue-dominant with R ~ G and B/R ~ 255/40: RGB::new(40, 40, 255) through
        color_multiply's g/b swap emits (40m, 40m, 255m) (calc/src/lib.rs:129-139) — what the oracle emits;
      * a quarter of the screenshot is black: `inside = false` renders the set's interior BLACK (calc/src/lib.rs:233) —
        what the oracle does with the same flag;
      * an oracle render of a deep view with -d shows the same two facts exactly."""
    import json
    import os

    here = os.path.dirname(os.path.abspath(__file__))
    st = json.load(open(os.path.join(here, "golden", "reference_screenshot_stats.json")))
    assert (st["width"], st["height"]) == (1000, 1000)
    want_ratio = 255.0 / 40.0
    assert abs(st["b_over_r_median"] - want_ratio) / want_ratio < 0.05, st          # 6.4 against 6.375, through a lossy codec
    assert st["b_over_r_p10"] > 5.0 and st["b_over_r_p90"] < 7.5, st
    assert abs(st["r_minus_g_median"]) <= 1.0 and st["abs_r_minus_g_p90"] <= 4.0, st  # R ~ G
    assert st["fraction_green_above_blue"] == 0.0, st                              # blue, never green
    assert st["black_fraction"] > 0.1, st                                          # -d: the interior is black
    # the oracle, same colours, inside off, a zoomed view with both interior and exterior
    cfg = O.cli_config(200, 200, O.MANDELBROT, iterations=500, pos=(-0.7436447860, 0.1318252536), scale=(3000.0, 3000.0), inside=False)
    img = O.get_image(cfg).astype(np.int64)
    z, iters = O.escape_rows(cfg)
    z = z.reshape(200, 200, 2)
    # "inside" in the reference's sense: the orbit ended within stable_limit of the origin (calc/src/lib.rs:216; a bounded
    # orbit that ends further out is coloured as outside, KAT-3)
    inside = (z[..., 0] ** 2 + z[..., 1] ** 2) <= cfg.stable_limit
    assert inside.any() and (~inside).any()
    assert (img[inside] == 0).all()                                                # interior: BLACK
    R, G, B = img[..., 0], img[..., 1], img[..., 2]
    assert (R == G).all()                                                          # 40 m and 40 m, truncated alike
    meas = (~inside) & (R >= 16) & (B < 250)
    ratio = B[meas] / R[meas]
    assert meas.sum() > 100 and abs(np.median(ratio) - want_ratio) / want_ratio < 0.03


def test_geometry_against_the_reference_s_own_screenshot():
    """the one output the reference holds, screenshots/mandelbrot-1000000x.avif, is examples.md:29's view
    (-s 500000 -x -.7436447860 -y .1318252536 -i 4000 -d) rendered square — found by searching scale x iterations with the
    oracle (tests/golden/make_screenshot_geometry.py, which committed the screenshot's BLACK MASK at 125 x 125 cells and the
    search table, not the image).  The oracle's interior mask must overlap it in the reference's orientation and in no
    other: that pins `x / height`, y-down, the centre convention and the meaning of --scale (calc/src/lib.rs:182-197) against
    something the REFERENCE produced.  (A lossy, rescaled image pins no bits: parity stays 'unpinned' in the task's sense.)"""
    import json
    import os

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_screenshot_geometry.json")
    doc = json.load(open(path))
    n = doc["mask_cells"]
    ref = np.array([[c == "1" for c in bin(int(row, 16))[2:].zfill(n)] for row in doc["mask_rows_hex"]])
    assert ref.shape == (n, n) and abs(ref.mean() - doc["black_fraction_screenshot"]) < 0.01
    v = doc["view"]
    assert (v["scale"], v["iterations"], v["inside"]) == (500000.0, 4000, 0)  # examples.md:29
    cfg = O.cli_config(500, 500, O.MANDELBROT, iterations=v["iterations"], scale=(v["scale"], v["scale"]),
                       pos=(float.fromhex(v["pos"][0]), float.fromhex(v["pos"][1])), inside=0)
    _, iters = O.escape_rows(cfg)
    black = iters >= cfg.iterations  # what `inside = false` paints BLACK (calc/src/lib.rs:233)
    # ... and the colour path agrees: those pixels (bar the bounded orbits that end with |z|^2 > stable_limit, which are
    # coloured as outside: KAT-3), and only those (bar a few dim exterior ones), come out (0, 0, 0)
    img = O.get_image(cfg)
    assert (img[black].max(axis=1) == 0).mean() > 0.97 and (img[~black].max(axis=1) > 0).mean() > 0.99
    m = black.reshape(n, 4, n, 4).mean(axis=(1, 3)) >= 0.5

    def iou(a, b):
        return (a & b).sum() / max((a | b).sum(), 1)

    got = {"identity": iou(m, ref), "flip_y": iou(m[::-1], ref), "flip_x": iou(m[:, ::-1], ref), "transpose": iou(m.T, ref),
           "rot180": iou(m[::-1, ::-1], ref)}
    assert got["identity"] > 0.93, got
    assert max(got["flip_y"], got["flip_x"], got["transpose"], got["rot180"]) < 0.6, got
    # the committed search: this view is the best of the grid, and brightness fits best near exposure / iterations ~ 1 / 1500
    best = max(doc["iou_search"], key=lambda r: r["identity"])
    assert (best["scale"], best["iterations"]) == (500000.0, 4000) and best["identity"] > 0.97
    assert all(r["identity"] < 0.8 for r in doc["iou_search"] if r["scale"] != 500000.0)
    fit = max(doc["colour_fit"], key=lambda r: r["identity"]["psnr_db"])
    assert fit["identity"]["psnr_db"] > 20 and fit["identity"]["corr_blue"] > 0.9
    assert all(fit[k]["psnr_db"] < fit["identity"]["psnr_db"] - 5 for k in ("flip_y", "flip_x", "transpose"))
