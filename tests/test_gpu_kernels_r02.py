"""Round-2 kernel work, through the C ABI on the GPU box, against the oracle:
  * contraction-sensitive known-answer tests (tests/exact_model.py) on every orbit-loop form,
  * the colour filter (f32 bracket of nu + exact fallback): its bracket scanned over EVERY f32, and byte
    identity with the always-exact path,
  * the work-queue kernel (persistent waves, LDS result stack): byte identity with the oracle on ragged,
    tiny, in-place / block-cyclic, palette, RGBA and mode-switching cases, and at BASELINE C4's full size,
  * the scaled loop's admissibility boundaries (limit 2^400, |c| at 2^-300 / 2^400; f32 2^+-30).
"""
import ctypes as C
import struct

import numpy as np
import pytest

import exact_model as M
import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fr():
    import torch  # noqa: F401  (first: see INTEGRATION.md §4)

    import fractal_renderer_amd

    assert fractal_renderer_amd.device_count() > 0, "no HIP device: the GPU tests need a real MI355X"
    fractal_renderer_amd.init(0)
    return fractal_renderer_amd


@pytest.fixture(scope="module")
def lib(fr):
    from fractal_renderer_amd import _native

    return _native.load()


def to_fr(fr, ocfg):
    return fr.Config.from_buffer_copy(bytes(ocfg))


def f32_bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


# ---- contraction-sensitive KATs on the device -----------------------------------------------------


def one_pixel_config(re, im, iterations, limit):
    """A 1x1 image whose only pixel is EXACTLY (re, im): coordinate = (0/1 - 0.5)/1 + pos with
    pos = v + 0.5, both steps exact for the KAT inputs (find_contraction_kats.py picked them so)."""
    return O.cli_config(1, 1, iterations=iterations, limit=limit, pos=(re + 0.5, im + 0.5), scale=(1.0, 1.0))


@pytest.mark.parametrize("fmt,kats", [("f64", M.CONTRACTION_KATS_F64), ("f32", M.CONTRACTION_KATS_F32)])
def test_contraction_sensitive_kats_on_every_loop_form(fr, lib, fmt, kats):
    """fl(fl(re*re) - fl(im*im)) etc. exactly as the reference rounds them (calc/src/lib.rs:88-89, 95,
    103-104): expected values from exact rational arithmetic, on the unscaled loop, both scaled loops
    (whose single fma is exact by construction) and the batch entry point."""
    from fractal_renderer_amd import _native

    prec = fr.Precision.F32 if fmt == "f32" else fr.Precision.F64
    for n, re, im, limit in kats:
        want = M.recursive(n, (re, im), (re, im), limit, fmt)
        pos, it = fr.recursive_batch(n, [(re, im)], [(re, im)], limit, prec)
        assert (pos[0, 0], pos[0, 1], int(it[0])) == (want[0][0], want[0][1], want[1]), ("batch", n, re, im)
        ocfg = one_pixel_config(re, im, n, limit)
        assert O.lib().fro_xy_to_imaginary(C.byref(ocfg), 0, 0).re == re
        assert O.lib().fro_xy_to_imaginary(C.byref(ocfg), 0, 0).im == im
        cfg = to_fr(fr, ocfg)
        for mode in (0, 2, 4):
            for tile in (0, 9, 10, 11, 808):
                try:
                    _native.check(lib.fr_set_loop_mode(mode))
                    _native.check(lib.fr_set_tile(tile))
                    z, iters = fr.escape_rows(cfg, 0, 1, prec)
                finally:
                    lib.fr_set_loop_mode(-1)
                    lib.fr_set_tile(0)
                assert (z[0, 0, 0], z[0, 0, 1], int(iters[0, 0])) == (want[0][0], want[0][1], want[1]), (mode, tile, n, re, im)


# ---- the colour filter ---------------------------------------------------------------------------


def test_colour_filter_bracket_holds_for_every_f32(fr, lib):
    """The filter brackets nu = log2(log2(sqrt(dist))/2) by its f32 estimate +- 2^-18.  Scan EVERY f32
    value of (float)dist in [2, 2^120] on the device: the estimate is within 1.5e-6 of the f64 path's nu
    (which leaves > 2x margin for the f64->f32 conversion of dist, 1.3e-7, and everything else)."""
    from fractal_renderer_amd import _native

    x = np.array([float(f32_bits(2.0)), float(f32_bits(2.0 ** 120))])
    y = np.zeros(2)
    _native.check(lib.fr_debug_math(4, x.ctypes.data, y.ctypes.data, 2))
    assert 0.0 < y[0] < 1.5e-6, y[0]
    assert y[0] + 1.3e-7 + 1e-12 < 2.0 ** -18


FILTER_CASES = [
    dict(width=640, height=480, iterations=200),
    dict(width=333, height=257, iterations=3, exposure=50.0),
    dict(width=333, height=257, iterations=1),
    dict(width=400, height=300, iterations=1024, exposure=255.0, primary_color=(255, 255, 255)),
    dict(width=400, height=300, iterations=50, exposure=-1.0),
    dict(width=400, height=300, iterations=50, exposure=1e-3),
    dict(width=400, height=300, iterations=77, stable_limit=0.5),   # dist in (0.5, 2): outside the filter's range
    dict(width=400, height=300, iterations=77, stable_limit=0.0, limit=2.0),
    dict(width=400, height=300, iterations=60, limit=1.5, stable_limit=30.0),
    dict(width=400, height=300, iterations=90, limit=1e30),          # dist up to 1e60+: beyond 2^120 -> exact path
    dict(width=512, height=512, iterations=300, scale=(1e6, 1e6), pos=(-0.7436447860, 0.1318252536)),
    dict(width=400, height=300, iterations=120, algo=O.JULIA, julia_set=(-0.8, 0.156)),
    dict(width=400, height=300, iterations=40, exposure=float("inf")),   # filter must switch itself off
    dict(width=400, height=300, iterations=40, exposure=float("nan")),
    dict(width=200, height=100, iterations=0),
    dict(width=400, height=300, iterations=300, exposure=1e-20),      # |K| below the f32 stage's range
    dict(width=400, height=300, iterations=300, exposure=1e25),       # ... and above it
    dict(width=400, height=300, iterations=2 ** 24 + 5, limit=2.0, scale=(0.05, 0.05), pos=(3.0, 3.0)),  # cap >= 2^24: escapes at once
    dict(width=640, height=480, iterations=16, exposure=255.0, primary_color=(255, 255, 255)),  # steep: many near-boundary values
]


@pytest.mark.parametrize("case", FILTER_CASES)
def test_colour_filter_gives_the_exact_paths_bytes(fr, case):
    kw = dict(case)
    w, h, algo = kw.pop("width"), kw.pop("height"), kw.pop("algo", O.MANDELBROT)
    ocfg = O.cli_config(w, h, algo, **kw)
    cfg = to_fr(fr, ocfg)
    want = O.get_image(ocfg)  # libm log2: what the reference calls
    for prec in (fr.Precision.F64, fr.Precision.F32):
        w32 = want if prec == fr.Precision.F64 else O.get_image(ocfg, O.F32)
        on = fr.get_image_rows(cfg, 0, h, prec, opts=fr.RenderOpts(colour_filter=1))   # f32 stage, then f64 stage
        f64_only = fr.get_image_rows(cfg, 0, h, prec, opts=fr.RenderOpts(colour_filter=2))
        off = fr.get_image_rows(cfg, 0, h, prec, opts=fr.RenderOpts(colour_filter=0))  # always the software log2
        assert np.array_equal(on, off) and np.array_equal(f64_only, off), (case, prec)
        assert np.array_equal(on, w32), (case, prec)


def test_colour_filter_full_size_c2_identical(fr, lib):
    """BASELINE C2 (16384^2): the filter's three settings (f32 + f64 stages, f64 stage only, off) give the same
    805 306 368 bytes."""
    import torch
    from fractal_renderer_amd import _native

    cfg = to_fr(fr, O.cli_config(16384, 16384, iterations=1024))
    need = 3 * 16384 * 16384
    s = torch.cuda.current_stream()
    imgs = []
    for flt in (1, 2, 0):
        d = torch.empty(need, dtype=torch.uint8, device="cuda:0")
        o = fr.RenderOpts(colour_filter=flt)
        _native.check(lib.fr_render_rows_rgb8_device_opts(C.byref(cfg), 0, 0, 16384, d.data_ptr(), need, s.cuda_stream,
                                                          C.byref(o)))
        imgs.append(d)
    torch.cuda.synchronize()
    assert torch.equal(imgs[0], imgs[2]) and torch.equal(imgs[1], imgs[2])


# ---- the work-queue kernel (tile = 10) --------------------------------------------------------------

QUEUE_CASES = [
    dict(width=1237, height=1001, iterations=300, algo=O.JULIA, julia_set=(-0.8, 0.156)),       # ragged both ways
    dict(width=64, height=32, iterations=100, algo=O.JULIA, julia_set=(-0.8, 0.156)),           # exactly one patch
    dict(width=65, height=33, iterations=100, algo=O.JULIA, julia_set=(0.285, 0.01)),           # one px over in both
    dict(width=64, height=16, iterations=100, algo=O.JULIA, julia_set=(-0.8, 0.156)),           # half a patch
    dict(width=3, height=2, iterations=50),                                                     # smaller than a patch row
    dict(width=700, height=520, iterations=400),                                                # Mandelbrot: lanes hit the cap
    dict(width=700, height=520, iterations=400, smooth=0),                                      # palette in dynamic LDS
    dict(width=700, height=520, iterations=2000, smooth=0, inside=0),                           # palette too large: per pixel
    dict(width=512, height=512, iterations=1, algo=O.JULIA, julia_set=(-0.8, 0.156)),
    dict(width=512, height=512, iterations=0),
    dict(width=800, height=600, iterations=250, algo=O.JULIA, julia_set=(0.0, 0.0)),            # c = 0: never admissible
    dict(width=800, height=600, iterations=250, algo=O.JULIA, julia_set=(-0.8, 0.156), pos=(0.3, -0.2), scale=(3.0, 0.7)),
    dict(width=1024, height=512, iterations=5000, algo=O.JULIA, julia_set=(-0.4, 0.6)),         # connected set: long orbits
    dict(width=900, height=700, iterations=300, limit=2.0),                                     # loop plan falls back
    dict(width=640, height=480, iterations=120, algo=O.BARNSLEY_FERN),                          # BLACK (calc/src/lib.rs:211)
]


@pytest.mark.parametrize("case", QUEUE_CASES)
def test_work_queue_kernel_matches_the_oracle(fr, case):
    kw = dict(case)
    w, h, algo = kw.pop("width"), kw.pop("height"), kw.pop("algo", O.MANDELBROT)
    ocfg = O.cli_config(w, h, algo, **kw)
    cfg = to_fr(fr, ocfg)
    for prec, oprec in ((fr.Precision.F64, O.F64), (fr.Precision.F32, O.F32)):
        want = O.get_image(ocfg, oprec)
        for loop_mode in (-1, 0, 2):
            got = fr.get_image_rows(cfg, 0, h, prec, opts=fr.RenderOpts(tile=10, loop_mode=loop_mode))
            assert np.array_equal(got, want), (case, prec, loop_mode)
        for minrun, quit16 in ((0, 1), (3, 16), (64, 4)):
            got = fr.get_image_rows(cfg, 0, h, prec, opts=fr.RenderOpts(tile=10, refill_minrun=minrun, refill_quit16=quit16))
            assert np.array_equal(got, want), (case, prec, minrun, quit16)


@pytest.mark.parametrize("case", QUEUE_CASES)
def test_two_pass_render_matches_the_oracle(fr, lib, case):
    """tile 11: strips to the end of their first episodes, the rest through the survivor lists (fr_kernels.hip,
    escape_first_kernel): every episode length / keep threshold, lists that overflow, loop plans that rule it out."""
    kw = dict(case)
    w, h, algo = kw.pop("width"), kw.pop("height"), kw.pop("algo", O.MANDELBROT)
    ocfg = O.cli_config(w, h, algo, **kw)
    cfg = to_fr(fr, ocfg)
    for prec, oprec in ((fr.Precision.F64, O.F64), (fr.Precision.F32, O.F32)):
        want = O.get_image(ocfg, oprec)
        for loop_mode in (-1, 0, 2):
            got = fr.get_image_rows(cfg, 0, h, prec, opts=fr.RenderOpts(tile=11, loop_mode=loop_mode))
            assert np.array_equal(got, want), (case, prec, loop_mode)
        # minrun = episode length, quit16 = lanes / 4 a tile must keep running to stay in the first pass
        for episode, keep16 in ((4, 1), (8, 16), (64, 12), (200, 8), (1000, 4)):
            got = fr.get_image_rows(cfg, 0, h, prec, opts=fr.RenderOpts(tile=11, refill_minrun=episode, refill_quit16=keep16))
            assert np.array_equal(got, want), (case, prec, episode, keep16)
        # tile 13: the first pass alone (no tile is handed over, no lists, no second kernel), same episode lengths
        for episode in (-1, 4, 8, 64, 200):
            got = fr.get_image_rows(cfg, 0, h, prec, opts=fr.RenderOpts(tile=13, refill_minrun=episode))
            assert np.array_equal(got, want), (case, prec, "first pass alone", episode)
        # the shortcut rules the two passes out (the launch takes the patch-refill kernel); no filter / no palette change their colour path
        for kw in (dict(cycle_shortcut=1), dict(colour_filter=0), dict(palette=0)):
            got = fr.get_image_rows(cfg, 0, h, prec, opts=fr.RenderOpts(tile=11, **kw))
            assert np.array_equal(got, want), (case, prec, kw)
        # the comparison variants: 12 = round 2's two kernels, 14 = round 2's second pass behind this round's first;
        # second-pass policies (through tile 10's numbers the second pass keeps its own defaults: RGBA and row bands below)
        for tile in (12, 14):
            for episode, keep16 in ((-1, -1), (8, 16), (200, 8)):
                got = fr.get_image_rows(cfg, 0, h, prec, opts=fr.RenderOpts(tile=tile, refill_minrun=episode, refill_quit16=keep16))
                assert np.array_equal(got, want), (case, prec, tile, episode, keep16)
        try:  # lists of 64 entries each: nearly everything overflows and is finished by the first pass itself
            lib.fr_debug_set_two_pass_capacity(64)
            for episode in (-1, 8):
                got = fr.get_image_rows(cfg, 0, h, prec, opts=fr.RenderOpts(tile=11, refill_minrun=episode))
                assert np.array_equal(got, want), (case, prec, "overflow", episode)
        finally:
            lib.fr_debug_set_two_pass_capacity(0)


def test_two_pass_render_from_concurrent_threads_and_through_the_host_path(fr, lib):
    """Five host threads, each on its own stream, render Julia images in two passes (tile 11: since round 4 the default
    dispatch takes them at this size only for views whose measured statistics call for them): they contend for the context's
    three survivor-list buffers and sixteen counter slots.  Then the host-buffer entry point, which renders the image in bands
    on two streams."""
    import threading

    import torch
    from fractal_renderer_amd import _native

    ocfgs = [O.cli_config(2048, 2048, O.JULIA, julia_set=js, iterations=it)
             for js, it in (((-0.8, 0.156), 600), ((0.285, 0.01), 512), ((-0.4, 0.6), 900), ((-0.8, 0.156), 700), ((0.001, 0.8), 520))]
    cfgs = [to_fr(fr, c) for c in ocfgs]
    precs = [0, 1, 0, 1, 0]
    want = []
    for cfg, prec in zip(cfgs, precs):  # the strip kernel, serially
        want.append(torch.from_numpy(fr.get_image_rows(cfg, 0, 2048, prec, opts=fr.RenderOpts(tile=8))).cuda().reshape(-1))
    errs = []

    def work(i):
        try:
            dev = torch.device("cuda", 0)
            stream = torch.cuda.Stream(dev)
            out = torch.empty(2048 * 2048 * 3, dtype=torch.uint8, device=dev)
            name = C.create_string_buffer(160)
            _native.check(lib.fr_set_profiling(1))
            for rep in range(12):
                with torch.cuda.stream(stream):
                    out.zero_()
                    o = fr.RenderOpts(tile=11)
                    _native.check(lib.fr_render_rows_rgb8_device_opts(C.byref(cfgs[i]), precs[i], 0, 2048, out.data_ptr(), out.numel(),
                                                                      stream.cuda_stream, C.byref(o)))
                stream.synchronize()
                _native.check(lib.fr_last_kernel_name(name, len(name)))
                if not name.value.startswith(b"escape_first_kernel"):
                    errs.append("thread %d: %r ran" % (i, name.value))
                    return
                if not torch.equal(out, want[i]):
                    errs.append("thread %d rep %d: mismatch" % (i, rep))
                    return
            _native.check(lib.fr_set_profiling(0))
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(cfgs))]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errs, errs
    # the host-buffer path: the 64 MiB bands of an 8192 x 5000 Julia image on alternating streams, each band two passes
    ocfg = O.cli_config(8192, 5000, O.JULIA, julia_set=(-0.8, 0.156), iterations=700)
    cfg = to_fr(fr, ocfg)
    img = fr.get_image(cfg)
    total, npx, sample = O.sample_image(ocfg, 8, 8)
    assert np.array_equal(img[::8, ::8], sample)
    assert np.array_equal(img, fr.get_image_rows(cfg, 0, 5000, 0, opts=fr.RenderOpts(tile=8)))


@pytest.mark.parametrize("tile", [10, 11, 14])
def test_work_queue_kernel_row_bands_rgba_and_in_place_blocks(fr, lib, tile):
    import torch
    from fractal_renderer_amd import _native

    ocfg = O.cli_config(777, 613, O.JULIA, julia_set=(-0.8, 0.156), iterations=350)
    cfg = to_fr(fr, ocfg)
    want = O.get_image(ocfg)
    o = fr.RenderOpts(tile=tile)
    # row bands rendered on their own
    for y0, y1 in ((0, 613), (100, 117), (600, 613), (5, 6)):
        assert np.array_equal(fr.get_image_rows(cfg, y0, y1, 0, opts=o), want[y0:y1])
    s = torch.cuda.current_stream()
    # RGBA8
    d = torch.zeros(613 * 777 * 4, dtype=torch.uint8, device="cuda:0")
    _native.check(lib.fr_render_rows_rgba8_device_opts(C.byref(cfg), 0, 0, 613, d.data_ptr(), d.numel(), s.cuda_stream,
                                                       C.byref(o)))
    rgba = d.cpu().numpy().reshape(613, 777, 4)
    assert np.array_equal(rgba[..., :3], want) and (rgba[..., 3] == 255).all()
    # block-cyclic shares rendered IN PLACE into one image (what a multi-GPU root does), 3 shares
    img = torch.zeros(613 * 777 * 3, dtype=torch.uint8, device="cuda:0")
    for r in range(3):
        rows = C.c_uint64(0)
        _native.check(lib.fr_render_block_cyclic_range_rgb8_device_opts(
            C.byref(cfg), 0, 16, r, 3, 0, 1, img.data_ptr(), img.numel(), s.cuda_stream, C.byref(rows), C.byref(o)))
    assert np.array_equal(img.cpu().numpy().reshape(613, 777, 3), want)
    # and packed
    for r in range(3):
        rows = C.c_uint64(0)
        nrows = lib.fr_block_cyclic_rows(613, 16, r, 3)
        part = torch.zeros(nrows * 777 * 3, dtype=torch.uint8, device="cuda:0")
        _native.check(lib.fr_render_block_cyclic_range_rgb8_device_opts(
            C.byref(cfg), 0, 16, r, 3, 0, 0, part.data_ptr(), part.numel(), s.cuda_stream, C.byref(rows), C.byref(o)))
        got = part.cpu().numpy().reshape(nrows, 777, 3)
        idx = [y for b in range(r, (613 + 15) // 16, 3) for y in range(b * 16, min(613, b * 16 + 16))]
        assert rows.value == nrows and np.array_equal(got, want[idx])


@pytest.mark.parametrize("prec_name", ["f32", "f64"])
def test_full_size_c4_work_queue_kernel(fr, lib, prec_name):
    """BASELINE C4 (Julia, 16384^2, 4096 iterations) through the work-queue kernel (tile 10): ALL 805 306 368 bytes
    against the oracle (libm log2; the whole image is 1.2e10 pixel-iterations, a few seconds of CPU — VERDICT r02 #3:
    no sampling where the full comparison is this cheap), the 180-degree symmetry of the Julia image, and byte identity
    with the default dispatch (two passes), the first pass alone, patch refill, and with the colour filter off."""
    import torch
    from fractal_renderer_amd import _native

    ocfg = O.cli_config(16384, 16384, O.JULIA, julia_set=(-0.8, 0.156), iterations=4096)
    cfg = to_fr(fr, ocfg)
    prec = 1 if prec_name == "f32" else 0
    need = 3 * 16384 * 16384
    s = torch.cuda.current_stream()

    def render(**kw):
        d = torch.empty(need, dtype=torch.uint8, device="cuda:0")
        o = fr.RenderOpts(**kw)
        _native.check(lib.fr_set_profiling(1))
        _native.check(lib.fr_render_rows_rgb8_device_opts(C.byref(cfg), prec, 0, 16384, d.data_ptr(), need, s.cuda_stream,
                                                          C.byref(o)))
        name = C.create_string_buffer(160)
        _native.check(lib.fr_last_kernel_name(name, len(name)))
        _native.check(lib.fr_set_profiling(0))
        return d, name.value

    img, name = render(tile=10)
    assert name.startswith(b"escape_queue_kernel"), name
    torch.cuda.synchronize()
    total, npx, want = O.sample_image(ocfg, 1, 1, O.F32 if prec else O.F64)  # the whole image on the CPU
    view = img.view(16384, 16384, 3)
    assert npx == 16384 * 16384
    got = view.cpu().numpy()
    assert np.array_equal(got, want)
    del got, want
    gpu_total, _ = fr.count_iterations(cfg, 0, 16384, 1, 1, prec)
    assert gpu_total == total  # the executed-iteration sum of the whole image, CPU vs device
    assert torch.equal(view[1:, 1:], torch.flip(view[1:, 1:], dims=(0, 1)))
    # the default dispatch for an image like this: two passes; the patch-refill kernel; both with the filter off
    for kw, kernel in ((dict(), b"escape_first_kernel + escape_second_kernel"), (dict(tile=9), b"escape_refill_kernel"), (dict(tile=13), b"escape_first_kernel<"),
                       (dict(tile=11, refill_minrun=128, refill_quit16=16), b"escape_first_kernel + escape_second_kernel"),
                       (dict(colour_filter=0), b"escape_first_kernel"), (dict(tile=10, colour_filter=0), b"escape_queue_kernel")):
        other, name_o = render(**kw)
        assert name_o.startswith(kernel), (kw, name_o)
        assert torch.equal(img, other), kw
        del other


@pytest.mark.parametrize("view", ["c2", "c3"])
def test_full_size_mandelbrot_work_queue_kernel_equals_default(fr, lib, view):
    """BASELINE C2 and C3 (16384^2; C3: zoom 10^6, 65536 iterations — lanes that run to the cap, patches that hold
    the im == 0 row) through the work-queue kernel: the same 805 306 368 bytes as the default strip kernel, which
    test_gpu_parity.py pins against the oracle at these sizes."""
    import torch
    from fractal_renderer_amd import _native

    if view == "c2":
        ocfg = O.cli_config(16384, 16384, iterations=1024)
    else:
        ocfg = O.cli_config(16384, 16384, iterations=65536, scale=(1e6, 1e6), pos=(-0.7436447860, 0.1318252536))
    cfg = to_fr(fr, ocfg)
    need = 3 * 16384 * 16384
    s = torch.cuda.current_stream()
    imgs = []
    for tile in (8, 10, 11):  # 8: the strip kernel by name (the default dispatch samples the view: C2 -> strips, C3 -> either)
        d = torch.empty(need, dtype=torch.uint8, device="cuda:0")
        o = fr.RenderOpts(tile=tile)
        _native.check(lib.fr_set_profiling(1))
        _native.check(lib.fr_render_rows_rgb8_device_opts(C.byref(cfg), 0, 0, 16384, d.data_ptr(), need, s.cuda_stream,
                                                          C.byref(o)))
        name = C.create_string_buffer(160)
        _native.check(lib.fr_last_kernel_name(name, len(name)))
        _native.check(lib.fr_set_profiling(0))
        assert name.value.startswith({8: b"escape_strip_kernel", 10: b"escape_queue_kernel", 11: b"escape_first_kernel"}[tile]), name.value
        imgs.append(d)
    torch.cuda.synchronize()
    assert torch.equal(imgs[0], imgs[1]) and torch.equal(imgs[0], imgs[2])


# ---- the scaled loop at the edges of its admissible range ---------------------------------------------


def _escape(fr, lib, ocfg, prec, mode):
    from fractal_renderer_amd import _native

    cfg = to_fr(fr, ocfg)
    try:
        _native.check(lib.fr_set_loop_mode(mode))
        return fr.escape_rows(cfg, 0, cfg.height, prec)
    finally:
        lib.fr_set_loop_mode(-1)


@pytest.mark.parametrize("limit", [2.0 ** 400, float(np.nextafter(2.0 ** 400, np.inf)), 2.0 ** 399, 2.0 ** 401])
def test_scaled_loop_limit_boundary_f64(fr, lib, limit):
    """The host admits the scaled loop only for |limit| <= 2^400 (f64); at the boundary and one ulp past
    it every loop form still gives the oracle's (z, iters) bit for bit."""
    ocfg = O.cli_config(96, 64, iterations=40, limit=limit)
    wz, wit = O.escape_rows(ocfg)
    for mode in (-1, 0, 2, 4):
        z, it = _escape(fr, lib, ocfg, fr.Precision.F64, mode)
        assert np.array_equal(it, wit) and np.array_equal(z.view(np.uint64), wz.view(np.uint64)), (limit, mode)


@pytest.mark.parametrize("limit", [2.0 ** 30, float(np.nextafter(np.float32(2.0 ** 30), np.float32(np.inf))), 2.0 ** 31])
def test_scaled_loop_limit_boundary_f32(fr, lib, limit):
    ocfg = O.cli_config(96, 64, iterations=40, limit=limit)
    wz, wit = O.escape_rows(ocfg, O.F32)
    for mode in (-1, 0, 2, 4):
        z, it = _escape(fr, lib, ocfg, fr.Precision.F32, mode)
        assert np.array_equal(it, wit) and np.array_equal(z.view(np.uint64), wz.view(np.uint64)), (limit, mode)


@pytest.mark.parametrize("mag", [2.0 ** -300, float(np.nextafter(2.0 ** -300, 0.0)), 2.0 ** -299, 2.0 ** 400,
                                 float(np.nextafter(2.0 ** 400, np.inf)), 2.0 ** -1000, 5e-324])
def test_scaled_loop_c_magnitude_boundary_f64(fr, lib, mag):
    """Lanes are admitted to the scaled loop only if every |c| component lies in [2^-300, 2^400]: Julia
    constants exactly at, one ulp inside and one ulp outside both ends (and deep in the subnormal
    range), on the scaled and unscaled forms, against the oracle."""
    for jset in ((mag, 0.3), (-0.4, mag), (mag, -mag)):
        ocfg = O.cli_config(80, 48, O.JULIA, julia_set=jset, iterations=30, limit=2.0 ** 390 if mag > 1 else 65536.0)
        wz, wit = O.escape_rows(ocfg)
        for mode in (-1, 0, 4):
            z, it = _escape(fr, lib, ocfg, fr.Precision.F64, mode)
            nan = np.isnan(wz)
            assert np.array_equal(it, wit), (mag, jset, mode)
            assert np.array_equal(nan, np.isnan(z)) and np.array_equal(z.view(np.uint64)[~nan], wz.view(np.uint64)[~nan])


@pytest.mark.parametrize("mag", [2.0 ** -30, float(np.nextafter(np.float32(2.0 ** -30), np.float32(0))), 2.0 ** 30,
                                 float(np.nextafter(np.float32(2.0 ** 30), np.float32(np.inf))), 2.0 ** -140])
def test_scaled_loop_c_magnitude_boundary_f32(fr, lib, mag):
    for jset in ((mag, 0.3), (-0.4, mag)):
        ocfg = O.cli_config(80, 48, O.JULIA, julia_set=jset, iterations=30, limit=2.0 ** 28 if mag > 1 else 65536.0)
        wz, wit = O.escape_rows(ocfg, O.F32)
        for mode in (-1, 0, 4):
            z, it = _escape(fr, lib, ocfg, fr.Precision.F32, mode)
            nan = np.isnan(wz)
            assert np.array_equal(it, wit), (mag, jset, mode)
            assert np.array_equal(nan, np.isnan(z)) and np.array_equal(z.view(np.uint64)[~nan], wz.view(np.uint64)[~nan])


def test_default_dispatch_chooses_the_kernel_from_a_sample_of_the_image(fr, lib):
    """for launches of 131 072 tiles and more the default dispatch renders a sample of 256 tiles and
    picks strips / the first pass alone / two passes from what it sees — for Mandelbrot views too.  Whatever it picks,
    the bytes are those of the strip kernel; the choice is remembered per view; fr_set_dispatch_sampling(0) restores the
    rule by algorithm.  (Device-pointer renders: the host-buffer path renders in bands, which are smaller launches.)"""
    import ctypes as C

    import torch

    from fractal_renderer_amd import _native

    w, h = 4096, 2048  # exactly 131 072 tiles: the blocking sample, the rule of launches under 8192 x 4096
    views = [
        # a dust: two passes in f64 (4-tile first pass), one-tile strips in f32
        ("julia dust", dict(algo=O.JULIA, iterations=4096, julia_set=(-0.8, 0.156)), (b"escape_first_kernel + escape_second_kernel<double, 4-tile",
                                                                                      b"escape_strip_kernel<float, 1 tile")),
        ("mandelbrot default", dict(algo=O.MANDELBROT, iterations=1024), (b"escape_strip_kernel<float, 1 tile", b"escape_strip_kernel<double, 1 tile")),
        # orbits of a dozen iterations everywhere: the first pass alone, 7-tile strips
        ("mandelbrot exterior", dict(algo=O.MANDELBROT, iterations=4096, pos=(-1.9, 0.15), scale=(4.0, 4.0)), b"escape_first_kernel<"),
        # short orbits too, but the first pass cannot run this constant: the strip kernel with 4-tile strips
        ("julia dendrite (c.re = 0: the scaled loop is not admissible)", dict(algo=O.JULIA, iterations=512, julia_set=(0.0, 1.0)),
         (b"escape_strip_kernel<float, 4 tiles", b"escape_strip_kernel<double, 4 tiles")),
    ]
    name = C.create_string_buffer(256)
    out = torch.empty(w * h * 3, dtype=torch.uint8, device="cuda")

    def render(cfg, prec, tile):
        o = fr.RenderOpts(tile=tile)
        _native.check(lib.fr_render_rows_rgb8_device_opts(C.byref(cfg), int(prec), 0, h, out.data_ptr(), out.numel(), None, C.byref(o)))
        torch.cuda.synchronize()
        _native.check(lib.fr_last_kernel_name(name, 256))
        return out.clone(), name.value

    try:
        _native.check(lib.fr_set_profiling(1))
        for label, kw, expect in views:
            algo = kw.pop("algo")
            ocfg = O.cli_config(w, h, algo, **kw)
            cfg = to_fr(fr, ocfg)
            st = (C.c_double * 8)()
            _native.check(lib.fr_debug_sample_view(C.byref(cfg), 1, st))
            assert st[2] == 256 and 0.0 < st[6] <= 1.0 and st[0] > 0, (label, list(st))
            for prec in (fr.Precision.F32, fr.Precision.F64):
                want, _ = render(cfg, prec, 8)
                for rep in range(2):  # the second render finds the view remembered
                    got, kname = render(cfg, prec, 0)
                    assert torch.equal(got, want), (label, prec, rep)
                    assert kname.startswith(expect), (label, prec, kname)
        # sampling off: two passes for Julia, strips for Mandelbrot, whatever the view
        _native.check(lib.fr_set_dispatch_sampling(0))
        ocfg = O.cli_config(w, h, O.MANDELBROT, iterations=4096, pos=(-1.9, 0.15), scale=(4.0, 4.0))
        cfg = to_fr(fr, ocfg)
        got, kname = render(cfg, fr.Precision.F32, 0)
        assert kname.startswith(b"escape_strip_kernel"), kname
        alone, kname = render(cfg, fr.Precision.F32, 13)
        assert kname.startswith(b"escape_first_kernel<"), kname
        assert torch.equal(got, alone)
    finally:
        _native.check(lib.fr_set_dispatch_sampling(1))
        _native.check(lib.fr_set_profiling(0))


@pytest.mark.parametrize("prec_name", ["f64", "f32"])
def test_view_sample_statistics_match_the_oracles_escape_counts(fr, lib, prec_name):
    """The sample the default dispatch decides on (view_sample_kernel) against the same statistics
    computed from the ORACLE's escape counts, exactly (they are integers): executed iterations, 64 x the tiles' longest
    orbits, tiles, lanes at the cap, and — replaying the first pass's episode schedule per tile — lanes handed over,
    lane-iterations idled by finishing them in place, iterations they still have to run."""
    from fractal_renderer_amd import _native

    oprec, prec = (O.F32, 1) if prec_name == "f32" else (O.F64, 0)
    for algo, kw in ((O.JULIA, dict(julia_set=(-0.8, 0.156), iterations=700)), (O.MANDELBROT, dict(iterations=300)),
                     (O.MANDELBROT, dict(iterations=5000, pos=(-0.7436, 0.1402), scale=(200.0, 200.0)))):
        w, h = 1031, 777  # ragged: tile origins are cell centres aligned down to 8
        ocfg = O.cli_config(w, h, algo, **kw)
        cfg = to_fr(fr, ocfg)
        st = (C.c_double * 8)()
        _native.check(lib.fr_debug_sample_view(C.byref(cfg), prec, st))
        _, iters = O.escape_rows(ocfg, oprec)
        iters = iters.reshape(h, w).astype(np.int64)
        cap = ocfg.iterations
        cap_s = min(cap, 4096)
        executed_all = np.minimum(np.where(iters < cap, iters + 1, cap), cap_s)  # the sample's loop stops at cap_s
        tot = dict(sum=0, mx=0, tiles=0, capped=0, handed=0, waste=0, rest=0)
        for tj in range(16):
            for ti in range(16):
                col0 = ((2 * ti + 1) * w // 32) & ~7
                row0 = ((2 * tj + 1) * h // 32) & ~7
                ex = np.zeros((8, 8), np.int64)
                blk = executed_all[row0:row0 + 8, col0:col0 + 8]
                ex[:blk.shape[0], :blk.shape[1]] = blk  # lanes past the image edge count 0
                valid = np.zeros((8, 8), bool)
                valid[:blk.shape[0], :blk.shape[1]] = True
                mx = int(ex.max())
                tot["sum"] += int(ex.sum()); tot["mx"] += 64 * mx; tot["tiles"] += 1
                tot["capped"] += int((valid & (ex == cap_s)).sum())
                e, ln, hands = 64, 64, False
                nrun = 0
                while e < mx:
                    nrun = int((ex > e).sum())
                    if nrun < 48:
                        hands = nrun > 0
                        break
                    if e >= 8 * 64 and ln < 16 * 64:
                        ln += ln
                    e += ln
                rest = int(np.maximum(ex - e, 0).sum())
                if hands:
                    tot["handed"] += nrun; tot["waste"] += 64 * (mx - e) - rest; tot["rest"] += rest
        got = dict(sum=st[0], mx=st[1], tiles=st[2], capped=st[3], handed=st[4], waste=st[5], rest=st[7])
        assert {k: int(v) for k, v in got.items()} == tot, (prec_name, algo, kw)
        assert st[6] == pytest.approx(tot["sum"] / tot["mx"])


def test_two_pass_kernels_past_four_gigabytes_of_output(fr, lib):
    """A device-pointer render whose image is larger than 4 GiB (36 000 x 40 000 x 3 = 4.32 GB): the first pass and the
    second pass then address pixels with 64-bit arithmetic instead of a scalar base + 32-bit lane offset (`narrow` in
    escape_first_kernel / escape_second_kernel) — a path no host-buffer render reaches (its bands are 64 MiB).  Two passes
    (11), the first pass alone (13) and round 2's second pass (14) against the strip kernel (8), on the device."""
    import torch
    from fractal_renderer_amd import _native

    w, h = 40000, 36000
    ocfg = O.cli_config(w, h, O.JULIA, julia_set=(-0.8, 0.156), iterations=120)
    cfg = to_fr(fr, ocfg)
    need = 3 * w * h
    assert need > 2 ** 32
    s = torch.cuda.current_stream()

    def render(tile):
        d = torch.empty(need, dtype=torch.uint8, device="cuda:0")
        o = fr.RenderOpts(tile=tile)
        _native.check(lib.fr_render_rows_rgb8_device_opts(C.byref(cfg), 0, 0, h, d.data_ptr(), need, s.cuda_stream, C.byref(o)))
        torch.cuda.synchronize()
        return d

    ref = render(8)
    # rows of the reference against the oracle: the first, one in the middle, the last (offsets past 2^32)
    got = ref.view(h, w, 3)
    for y in (0, h // 2 + 1, h - 1):
        assert np.array_equal(got[y].cpu().numpy(), O.get_image(ocfg, O.F64, y, y + 1)[0]), y
    for tile in (11, 13, 14):
        d = render(tile)
        assert torch.equal(d, ref), tile
        del d
    # RGBA8 (5.76 GB): one dword per pixel through the same 64-bit addressing
    d = torch.empty(4 * w * h, dtype=torch.uint8, device="cuda:0")
    o = fr.RenderOpts(tile=11)
    _native.check(lib.fr_render_rows_rgba8_device_opts(C.byref(cfg), 0, 0, h, d.data_ptr(), d.numel(), s.cuda_stream, C.byref(o)))
    torch.cuda.synchronize()
    rgba = d.view(h, w, 4)
    for y0 in range(0, h, 4000):  # in slabs: a strided comparison of the whole image would copy it
        assert torch.equal(rgba[y0:y0 + 4000, :, :3], got[y0:y0 + 4000]), y0
        assert bool((rgba[y0:y0 + 4000, :, 3] == 255).all()), y0
