"""The orbit-loop plan (fr_api.hip: plan_loop) is host arithmetic: pinned here without a device.

What must hold for the speculative long blocks (fr_kernels.hip: FR_SC_SPEC_BODY) to be exact is decided by the host:
whatever loop form a wave runs (the 4-iteration scaled loop, the first pass's blocks, the unscaled loop), they are on only
with 16 <= limit^2 <= 2^1000 (f32: 2^100), every |c| component <= limit^2 / 8 and finite starts: an orbit past the limit
then grows monotonically and passes the limit before anything overflows, so an escape inside a block is still visible at
its end and no orbit turns NaN without having escaped."""
import ctypes as C

import pytest

import oracle_lib as O


@pytest.fixture(scope="module")
def lib():
    from fractal_renderer_amd import _native

    return _native.load()


def plan(lib, ocfg, precision=0, mode=-1):
    import fractal_renderer_amd as fr
    from fractal_renderer_amd import _native

    cfg = fr.Config.from_buffer_copy(bytes(ocfg))
    lm, t, sq = C.c_uint32(), C.c_double(), C.c_uint32()
    try:
        _native.check(lib.fr_set_loop_mode(mode))
        _native.check(lib.fr_debug_loop_plan(C.byref(cfg), precision, C.byref(lm), C.byref(t), C.byref(sq)))
    finally:
        lib.fr_set_loop_mode(-1)
    return lm.value, t.value, sq.value


def test_default_view_speculates_after_sixteen_quiet_iterations(lib):
    for prec in (0, 1):
        lm, t, sq = plan(lib, O.cli_config(16384, 16384, iterations=1024), prec)
        assert lm == 4 and 6.0 < t < 7.5 and sq == 16, (prec, lm, t, sq)
        # T is the largest bound under which three more iterations cannot escape: g(g(g(T))) <= limit^2
        g = lambda d, cmax=1.85: 2.0 * (d + cmax) ** 2  # noqa: E731
        assert g(g(g(t))) <= 65536.0 ** 2


def test_selector_five_is_automatic_without_speculation(lib):
    ocfg = O.cli_config(1920, 1080, iterations=1024)
    assert plan(lib, ocfg, mode=5) == (4, plan(lib, ocfg)[1], 0)
    assert plan(lib, ocfg, mode=0) == (0, 0.0, 16)  # the unscaled loop speculates too
    lm, t, sq = plan(lib, ocfg, mode=2)
    assert lm == 2 and t > 6.8 and sq == 16  # (the two-iteration blocks themselves do not; a wave that falls back to the unscaled loop does)
    assert lib.fr_set_loop_mode(3) != 0 and lib.fr_set_loop_mode(6) != 0
    lib.fr_set_loop_mode(-1)


@pytest.mark.parametrize("limit,want_mode,want_spec", [
    (65536.0, 4, 16),
    (30000.0, 4, 16),   # T = 4.9
    (20000.0, 2, 16),   # four iterations would need T < 4.5: the two-iteration loop
    (4.5, 0, 16),       # nothing can be skipped: the unscaled loop, speculating (|c| <= 2.475 <= 20.25 / 8)
    (4.0, 0, 0),        # limit^2 = 16 would do, but this view's |c| reaches 2.475 > 16 / 8
    (3.99, 0, 0),
    (2.0, 0, 0),
    (2.0 ** 400, 4, 16),
    (2.0 ** 401, 0, 16),  # outside the range the scaled form is proven for; the unscaled loop still speculates
    (2.0 ** 500, 0, 16),  # limit^2 = 2^1000: the last that leaves room above it
    (2.0 ** 501, 0, 0),
    (2.0 ** 600, 0, 0),   # limit^2 = +inf: nothing ever escapes, an overflowing orbit turns NaN and must not be speculated on
    (float("nan"), 0, 0),
    (-65536.0, 4, 16),    # the reference squares the limit
])
def test_limits(lib, limit, want_mode, want_spec):
    lm, t, sq = plan(lib, O.cli_config(750, 500, iterations=200, limit=limit))
    assert (lm, sq) == (want_mode, want_spec), (limit, lm, t, sq)


def test_forced_scaled_loop_with_a_small_limit_still_needs_the_growth_conditions(lib):
    # forced 4: any positive T is taken; limit 1000 -> T = 0.9, limit^2 / 8 = 125 000 >= |c|: speculation allowed
    lm, t, sq = plan(lib, O.cli_config(320, 200, iterations=300, limit=1000.0), mode=4)
    assert lm == 4 and 0.0 < t < 1.5 and sq == 16
    # a view four units off the origin: |c| up to 5.9, T = 2.6 when forced — speculation allowed (5.9 <= limit^2 / 8)
    lm, t, sq = plan(lib, O.cli_config(320, 200, iterations=300, pos=(4.0, 0.0)), mode=4)
    assert lm == 4 and 0.0 < t < 4.5 and sq == 16
    assert plan(lib, O.cli_config(320, 200, iterations=300, pos=(4.0, 0.0)))[0] == 2  # (automatic: T < 4.5 -> two-iteration blocks)
    # |c| = 1e9: no T exists (T > 0 needs |c| < sqrt(limit^2 / 2), far inside the growth condition's limit^2 / 8)
    assert plan(lib, O.cli_config(320, 200, iterations=300, pos=(1e9, 0.0)), mode=4) == (0, 0.0, 0)  # and 1e9 > limit^2 / 8
    assert plan(lib, O.cli_config(320, 200, iterations=300, pos=(1e8, 0.0)), mode=4) == (0, 0.0, 16)  # 1e8 <= 5.4e8: unscaled, speculating


def test_f32_limits_and_non_finite_views(lib):
    assert plan(lib, O.cli_config(320, 200, iterations=300, limit=1e15), 1) == (0, 0.0, 16)  # (f32)limit^2 = 1e30 <= 2^100
    assert plan(lib, O.cli_config(320, 200, iterations=300, limit=1e16), 1) == (0, 0.0, 0)
    assert plan(lib, O.cli_config(320, 200, iterations=300, limit=1e30), 1) == (0, 0.0, 0)   # (f32)limit^2 = +inf
    for kw in (dict(scale=(0.0, 0.4)), dict(pos=(float("nan"), 0.0)), dict(pos=(float("inf"), 0.0)), dict(scale=(0.4, float("nan")))):
        for prec in (0, 1):
            assert plan(lib, O.cli_config(320, 200, iterations=300, **kw), prec) == (0, 0.0, 0), (kw, prec)
    # Julia: the starts (the pixel coordinates) must be finite too, not only the constant
    assert plan(lib, O.cli_config(320, 200, O.JULIA, iterations=300, julia_set=(-1.0, 0.0)))[2] == 16
    assert plan(lib, O.cli_config(320, 200, O.JULIA, iterations=300, julia_set=(-1.0, 0.0), scale=(0.0, 0.0)))[2] == 0
    assert plan(lib, O.cli_config(320, 200, O.JULIA, iterations=300, julia_set=(1e9, 0.0)))[2] == 0


def test_julia_uses_the_constant_c(lib):
    ocfg = O.cli_config(1024, 1024, O.JULIA, iterations=4096, julia_set=(-0.8, 0.156))
    lm, t, sq = plan(lib, ocfg, 1)
    assert lm == 4 and sq == 16 and 7.5 < t < 8.5  # |c| <= 0.8: a larger T than the Mandelbrot view's 6.8


def test_fern_and_null_arguments(lib):
    lm, t, sq = plan(lib, O.cli_config(100, 100, O.BARNSLEY_FERN))
    assert (lm, t, sq) == (0, 0.0, 0)
    assert lib.fr_debug_loop_plan(None, 0, None, None, None) != 0
