"""Speculative long blocks of the scaled orbit loop (fr_kernels.hip: FR_SC_SPEC_BODY, FR_FB_SPEC_ASM) against the CPU oracle.

A wave that has been quiet for 16 iterations runs 16 unchecked iterations at a time and tests once at their end; a lane that
escapes inside such a block must still be reported with the exact index and position recursive() returns
(calc/src/lib.rs:245-257).  The cases here are chosen so that escapes happen LATE, after long quiet stretches (deep boundary
views), at every remainder of the iteration cap modulo 4 and 16, through every kernel that carries the blocks (strips,
the refilling kernel, the first pass's later episodes in both counting forms), in both precisions, with speculation on
(-1 / 4) and off (5)."""
import numpy as np
import pytest

import oracle_lib as O
from test_gpu_parity import fr, oracle_image, same_f64, to_fr  # noqa: F401  (fixture + helpers)

pytestmark = pytest.mark.gpu

SEAHORSE = dict(pos=(-0.7436447860, 0.1318252536), scale=(500.0, 500.0))
ELEPHANT = dict(pos=(0.2925, 0.0149), scale=(60.0, 60.0))


def check(fr, ocfg, modes=(-1, 5), tiles=(0,), precisions=("f64", "f32")):
    from fractal_renderer_amd import _native

    lib = _native.load()
    cfg = to_fr(fr, ocfg)
    try:
        for pn in precisions:
            op, fp = (O.F64, fr.Precision.F64) if pn == "f64" else (O.F32, fr.Precision.F32)
            wz, wit = O.escape_rows(ocfg, op)
            wimg = oracle_image(ocfg, op)
            for mode in modes:
                _native.check(lib.fr_set_loop_mode(mode))
                z, it = fr.escape_rows(cfg, precision=fp)
                assert np.array_equal(it, wit), (pn, mode, "escape indices")
                assert same_f64(z, wz), (pn, mode, "final positions")
                for tile in tiles:
                    got = fr.get_image_rows(cfg, 0, cfg.height, fp, opts=fr.RenderOpts(tile=tile, loop_mode=mode))
                    assert np.array_equal(got, wimg), (pn, mode, tile)
    finally:
        lib.fr_set_loop_mode(-1)


@pytest.mark.parametrize("iterations", [16, 17, 31, 32, 33, 47, 49, 250, 1000, 1023, 1037])
def test_every_remainder_of_the_cap(fr, iterations):
    """n mod 4 runs checked first; fewer than 16 iterations before the cap end the speculation: every combination."""
    check(fr, O.cli_config(200, 120, iterations=iterations, **SEAHORSE))


@pytest.mark.parametrize("view", ["seahorse", "elephant", "default"])
def test_late_escapes_through_every_kernel_that_speculates(fr, view):
    """Orbits that leave after hundreds of quiet iterations: the rollback path, in strips (0, 1-, 4-, 7-tile), the patch
    refill kernel (9), the first pass alone (13, 16) and with its second pass (11, 15)."""
    kw = dict(seahorse=SEAHORSE, elephant=ELEPHANT, default={})[view]
    check(fr, O.cli_config(328, 200, iterations=3000 if view != "default" else 700, **kw),
          modes=(-1, 4, 5), tiles=(0, 1, 4, 9, 13, 16, 11, 15))


def test_forced_scaled_loop_with_a_small_limit(fr):
    """limit 1000 forced through the 4-iteration loop: T = 0.9 — lanes wander above T all the time (rollbacks without an
    escape), and limit^2 = 1e6 is reached within two or three iterations of leaving."""
    check(fr, O.cli_config(200, 120, iterations=400, limit=1000.0), modes=(4, 5, 0), tiles=(0, 9, 13))


def test_julia_views(fr):
    """C4's dust (nothing stays: the speculation never pays, it must not cost correctness either) and a filled Julia set
    whose interior runs to the cap in speculative blocks."""
    check(fr, O.cli_config(256, 160, O.JULIA, iterations=2100, julia_set=(-0.8, 0.156)), tiles=(0, 11, 13))
    check(fr, O.cli_config(256, 160, O.JULIA, iterations=1500, julia_set=(-1.0, 1e-9)), tiles=(0, 11, 13, 9))
    check(fr, O.cli_config(256, 160, O.JULIA, iterations=900, julia_set=(0.285, 0.01)), tiles=(0, 11, 13))


@pytest.mark.parametrize("c", [(-1.0, 0.0), (0.0, 1.0), (0.25, 0.0), (-0.75, 0.0)])
def test_named_julia_sets_run_the_unscaled_loop_speculatively(fr, c):
    """A Julia constant with a zero component is outside the scaled form's proof (every wave takes the loop as written,
    FR_ORBIT_ASM): its interior runs in speculative blocks of 7-instruction iterations, its boundary escapes late."""
    check(fr, O.cli_config(256, 160, O.JULIA, iterations=1200, julia_set=c), modes=(-1, 0, 5), tiles=(0, 9, 13))


def test_unscaled_loop_forced_and_small_limits(fr):
    """Selector 0 on a deep boundary view; limits that leave no room for skipped checks (4.5: limit^2 / 8 just above the
    view's largest |c|; 3.9: below 16, no speculation) at caps with every remainder modulo 16."""
    check(fr, O.cli_config(200, 120, iterations=2500, **SEAHORSE), modes=(0,), tiles=(0, 9))
    for limit in (4.5, 3.9, 10.0, 150.0):
        for iterations in (333, 1030):
            check(fr, O.cli_config(200, 120, iterations=iterations, limit=limit), modes=(-1, 5), tiles=(0, 9))


def test_the_real_axis_strip(fr):
    """Mandelbrot rows through im = 0 are unscalable (a zero component of c): those strips run the unscaled loop inside a
    launch planned for the scaled one — both speculate; odd height puts the axis on a pixel row."""
    check(fr, O.cli_config(300, 201, iterations=900), tiles=(0, 9, 13))


def test_a_start_that_overflows_inside_a_block(fr):
    """limit 2^400: an orbit past it overflows to +inf and then NaN within a few iterations — inside one speculative block;
    the end test is `NOT (T >= dist)`, true for NaN, so the block is rolled back and the escape found at its iteration.
    (f64 only: (f32)limit^2 is +inf and nothing escapes, covered by the goldens.)"""
    check(fr, O.cli_config(160, 96, iterations=120, limit=2.0 ** 400, scale=(1e-3, 1e-3), pos=(0.0, 0.0)),
          modes=(-1, 5, 0), tiles=(0, 13), precisions=("f64",))
    check(fr, O.cli_config(160, 96, iterations=400, limit=2.0 ** 400), modes=(-1, 5), tiles=(0, 13), precisions=("f64",))
    # limit 2^500 (unscaled loop; limit^2 = 2^1000 is the largest that speculates) and 2^501 (does not)
    for limit in (2.0 ** 500, 2.0 ** 501):
        check(fr, O.cli_config(160, 96, iterations=200, limit=limit, scale=(1e-140, 1e-140), pos=(0.0, 0.0)),
              modes=(-1, 5), tiles=(0,), precisions=("f64",))


def test_same_device_bytes_with_and_without_speculation_at_4k(fr):
    """A GUI-sized frame (3840 x 2160, the zoomed view), device to device: identical bytes with speculation on and off."""
    import ctypes as C

    import torch

    from fractal_renderer_amd import _native

    lib = _native.load()
    cfg = to_fr(fr, O.cli_config(3840, 2160, iterations=2048, **SEAHORSE))
    outs = []
    try:
        for mode in (-1, 5):
            _native.check(lib.fr_set_loop_mode(mode))
            for prec in (fr.Precision.F64, fr.Precision.F32):
                d = torch.empty(3840 * 2160 * 3, dtype=torch.uint8, device="cuda")
                _native.check(lib.fr_render_rows_rgb8_device(C.byref(cfg), int(prec), 0, 2160, C.c_void_p(d.data_ptr()), d.numel(), None))
                outs.append(d)
    finally:
        lib.fr_set_loop_mode(-1)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[3])


def test_the_first_pass_runs_its_plain_form_where_nothing_stays(fr):
    """The view's statistics decide which form of the first-pass kernel a large launch gets (fr_api.hip: decide_from_sample):
    C4's dust — no sampled pixel at the cap, mean 44 — the plain one (the speculative form's set-up cost it 1.5 %), a deep
    boundary view the one whose later episodes speculate; selector 5 always the plain one.  The kernel's name says which."""
    import ctypes as C

    import torch

    from fractal_renderer_amd import _native

    lib = _native.load()
    w, h = 8192, 4096
    d = torch.empty(w * h * 3, dtype=torch.uint8, device="cuda")
    name = C.create_string_buffer(256)
    views = {
        "dust": (O.cli_config(w, h, O.JULIA, iterations=4096, julia_set=(-0.8, 0.156)), False),
        "deep": (O.cli_config(w, h, iterations=4096, pos=(-0.7436447860, 0.1318252536), scale=(1e6, 1e6)), True),
    }
    try:
        _native.check(lib.fr_set_profiling(1))
        for key, (ocfg, want_spec) in views.items():
            cfg = to_fr(fr, ocfg)
            for mode in (-1, 5):
                _native.check(lib.fr_set_loop_mode(mode))
                _native.check(lib.fr_render_rows_rgb8_device(C.byref(cfg), int(fr.Precision.F64), 0, h, C.c_void_p(d.data_ptr()), d.numel(), None))
                torch.cuda.synchronize()
                _native.check(lib.fr_last_kernel_name(name, len(name)))
                assert name.value.startswith(b"escape_first_kernel"), (key, mode, name.value)
                assert (b"speculative" in name.value) == (want_spec and mode == -1), (key, mode, name.value)
    finally:
        lib.fr_set_loop_mode(-1)
        lib.fr_set_profiling(0)
