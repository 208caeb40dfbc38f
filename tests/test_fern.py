"""Algo::BarnsleyFern (src/lib.rs:271-319, 369-401, 417-463): the oracle's restatement on CPU (hand-derived
known answers), and — on the GPU box — the device chaos game against it: BIT-exact with the same RNG
and walker split, statistically a single sequential orbit (the reference's shape).

The reference seeds its RNG from entropy (src/lib.rs:428), so its own output is only a sample; the RNG
here (Philox4x32-10 keyed by a seed) is build-defined and documented as such.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O


def fern_cfg(width=400, height=300, iterations=200_000, **kw):
    """Config::new(Algo::BarnsleyFern) (calc/src/lib.rs:39-69): primary new(4, 100, 3), secondary
    new(240, 240, 240), color_weight 0.01, scale 0.4, pos 0."""
    return O.config_new(O.BARNSLEY_FERN, width=width, height=height, iterations=iterations, **kw)


def subtract(pix, value, amount):
    """One Image::subtract_pixel (src/lib.rs:383-401) on stored fields (r, g, b), by hand: each channel
    c -> (c * 1.0 / (((1.0 / (v / 255.0)) - 1.0) * amount + 1.0)) as u8, then RGB::new(r', g', b') — whose
    second parameter is BLUE (calc/src/lib.rs:129-131) — stores {r: r', g: b', b: g'}."""
    def f(c, v):
        with np.errstate(divide="ignore"):
            d = (np.float64(1.0) / (np.float64(v) / 255.0) - 1.0) * amount + 1.0
            q = np.float64(c) * 1.0 / d
        return 0 if not q > 0 else 255 if q >= 255 else int(q)
    r, g, b = f(pix[0], value[0]), f(pix[1], value[1]), f(pix[2], value[2])
    return (r, b, g)


def test_single_point_lands_where_the_source_says():
    """iterations = 1: exactly one plotted point, the start (pos.re * width, pos.im * height) = (0, 0):
    x = (0 - 0) * esx + 400 / 2 = 200;  y = 300 - ((0 + (0 - 5) - 0.5) * esy + 150) with
    esy = 37 * 0.4 * 300 * 0.006 = 26.64 -> 300 - (-146.52 + 150) = 296.52 -> 296 (src/lib.rs:433-440)."""
    cfg = fern_cfg(iterations=1)
    img = O.fern_image(cfg, 1, 0, 1)
    stored_primary = (4, 3, 100)       # RGB::new(4, 100, 3) -> {r: 4, g: 3, b: 100}
    want = subtract((240, 240, 240), stored_primary, 0.01)
    # by hand: r' = 240 / ((255/4 - 1) * 0.01 + 1) = 240 / 1.6275 = 147.4 -> 147;  g' = 240 / ((255/3 - 1) * 0.01 + 1)
    # = 240 / 1.84 = 130.4 -> 130;  b' = 240 / ((255/100 - 1) * 0.01 + 1) = 240 / 1.0155 = 236.3 -> 236;
    # RGB::new(r', g', b') stores {r: r', g: b', b: g'}
    assert want == (147, 236, 130)
    hit = np.argwhere((img != 240).any(axis=2))
    assert hit.tolist() == [[296, 200]]
    assert tuple(img[296, 200]) == want
    # iterations / threads: 8 threads leave 0 points of 1 iteration -> untouched secondary colour
    assert (O.fern_image(cfg, 8, 0, 1) == 240).all()


def test_hits_on_one_pixel_compose_and_swap_green_and_blue():
    """pos = (0, 0) and the r < 0.01 map sends (x, y) to (0, 0.16 y): with scale 0 every point of the orbit
    plots onto one pixel — so m points = F applied m times, g and b trading places every time."""
    cfg = fern_cfg(width=40, height=30, iterations=5, scale=(0.0, 0.0))
    img = O.fern_image(cfg, 1, 3, 1)
    hit = np.argwhere((img != 240).any(axis=2))
    assert len(hit) == 1
    v = (240, 240, 240)
    for _ in range(5):
        v = subtract(v, (4, 3, 100), 0.01)
    assert tuple(img[hit[0][0], hit[0][1]]) == v


def test_oracle_fern_is_deterministic_and_seeded():
    cfg = fern_cfg()
    a = O.fern_image(cfg, 2, 7, 1)
    assert np.array_equal(a, O.fern_image(cfg, 2, 7, 1))
    assert not np.array_equal(a, O.fern_image(cfg, 2, 8, 1))
    # x > width is rejected, x == width spills into the next row; far-off points are dropped, not wrapped
    wide = fern_cfg(width=50, height=300, iterations=50_000, scale=(2.0, 0.4))
    img = O.fern_image(wide, 1, 1, 1)
    assert img.shape == (300, 50, 3)


def block_means(img, by=20, bx=20):
    h, w = img.shape[0] // by * by, img.shape[1] // bx * bx
    return img[:h, :w].astype(np.float64).reshape(h // by, by, w // bx, bx, 3).mean(axis=(1, 3))


def test_parallel_walkers_are_statistically_one_orbit_on_cpu():
    """The split into independently played pieces (what the GPU runs) against the single sequential orbit,
    both on the CPU restatement: block-averaged images agree as well as two sequential runs with different
    seeds agree with each other."""
    cfg = fern_cfg(iterations=1_500_000)
    seq_a, seq_b = O.fern_image(cfg, 1, 11, 1), O.fern_image(cfg, 1, 12, 1)
    par = O.fern_image(cfg, 1, 13, 4096)
    noise = np.abs(block_means(seq_a) - block_means(seq_b)).max()
    diff = np.abs(block_means(par) - block_means(seq_a)).max()
    assert noise > 0 and diff < 2.0 * noise + 0.5, (diff, noise)


# ---- the device ---------------------------------------------------------------------------------


@pytest.fixture(scope="module")
def fr():
    import torch  # noqa: F401

    import fractal_renderer_amd

    assert fractal_renderer_amd.device_count() > 0, "no HIP device: the GPU tests need a real MI355X"
    fractal_renderer_amd.init(0)
    return fractal_renderer_amd


GPU_CASES = [
    dict(), dict(width=333, height=517, iterations=300_001), dict(iterations=1), dict(iterations=0),
    dict(width=2000, height=1000, iterations=10_000_000),                      # Config::new(fern)'s own size and count
    dict(pos=(0.3, -0.2), scale=(0.7, 0.25), color_weight=0.2),
    dict(primary_color=(255, 0, 17), secondary_color=(9, 200, 255), color_weight=0.5),   # 255: untouched; 0: to black
    dict(width=50, height=300, scale=(2.0, 0.4)),                              # most points fall outside
    dict(scale=(0.0, 0.0), iterations=70_000),                                 # every point on one pixel: long F^m chain
    dict(color_weight=0.0), dict(color_weight=-0.5), dict(scale=(float("nan"), 0.4)),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", GPU_CASES)
def test_device_fern_is_bit_identical_to_the_restatement(fr, case):
    ocfg = fern_cfg(**case)
    cfg = fr.Config.from_buffer_copy(bytes(ocfg))
    for threads, seed, walkers in ((1, 0, 1), (1, 5, 7), (16, 2 ** 40 + 3, 1000), (3, 99, 262144)):
        want = O.fern_image(ocfg, threads, seed, walkers)
        got = fr.get_image_fern(cfg, threads, seed, walkers)
        assert np.array_equal(got, want), (case, threads, seed, walkers)


@pytest.mark.gpu
def test_device_fern_default_walkers_match_a_sequential_orbit_statistically(fr):
    ocfg = fern_cfg(width=2000, height=1000, iterations=10_000_000)
    cfg = fr.Config.from_buffer_copy(bytes(ocfg))
    seq_a, seq_b = O.fern_image(ocfg, 1, 21, 1), O.fern_image(ocfg, 1, 22, 1)
    gpu = fr.get_image_fern(cfg, 1, 23)          # walkers = 0: the library's own choice
    noise = np.abs(block_means(seq_a, 40, 40) - block_means(seq_b, 40, 40)).max()
    diff = np.abs(block_means(gpu, 40, 40) - block_means(seq_a, 40, 40)).max()
    assert noise > 0 and diff < 2.0 * noise + 0.5, (diff, noise)
    # the same fraction of the image is touched
    cover = lambda im: (im != 240).any(axis=2).mean()  # noqa: E731
    assert abs(cover(gpu) - cover(seq_a)) < 0.01


@pytest.mark.gpu
def test_device_fern_argument_errors_and_per_pixel_path(fr):
    from fractal_renderer_amd import _native

    lib = _native.load()
    cfg = fr.Config.from_buffer_copy(bytes(fern_cfg()))
    out = np.empty((300, 400, 3), dtype=np.uint8)
    assert lib.fr_render_fern_rgb8(C.byref(cfg), 0, 0, 0, out.ctypes.data, out.nbytes) == _native.FR_ERR_INVALID_ARGUMENT
    assert lib.fr_render_fern_rgb8(C.byref(cfg), 1, 0, 0, out.ctypes.data, 5) == _native.FR_ERR_BUFFER_TOO_SMALL
    assert lib.fr_render_fern_rgb8(C.byref(cfg), 1, 0, 1 << 20, out.ctypes.data, out.nbytes) == _native.FR_ERR_INVALID_ARGUMENT
    assert lib.fr_render_fern_rgb8(None, 1, 0, 0, out.ctypes.data, out.nbytes) == _native.FR_ERR_INVALID_ARGUMENT
    # calc::get_recursive_pixel renders BarnsleyFern BLACK (calc/src/lib.rs:211): the per-pixel path still does
    assert not fr.get_image(cfg).any()
