"""The default dispatch for GUI-sized frames: every frame src/gui.rs:56-82 asks for is below the size at which
the view sample may block, so the library samples WITHOUT blocking — the first frame of a view is dispatched by size, the
sample runs behind its render on a stream of the library's own, and the next frame of the same view (what every slider move
re-requests) is dispatched from the measured statistics.  Same bytes whatever is chosen."""
import ctypes as C
import time

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fr():
    import fractal_renderer_amd as fr

    fr.init(0)
    return fr


@pytest.fixture(scope="module")
def lib(fr):
    from fractal_renderer_amd import _native

    return _native.load()


def to_fr(fr, ocfg):
    return fr.Config.from_buffer_copy(bytes(ocfg))


def view_state(lib, cfg, prec, h):
    from fractal_renderer_amd import _native

    st, ch, k = C.c_int(-9), C.c_int(-9), C.c_uint32(99)
    _native.check(lib.fr_debug_view_choice(C.byref(cfg), int(prec), 0, h, C.byref(st), C.byref(ch), C.byref(k)))
    return st.value, ch.value, k.value


def wait_for_totals(lib, cfg, prec, h):
    for _ in range(2000):
        if view_state(lib, cfg, prec, h)[0] != 1:
            return
        time.sleep(0.001)
    raise AssertionError("the non-blocking sample never delivered its totals")


@pytest.mark.parametrize("prec_name", ["f32", "f64"])
def test_second_frame_of_a_gui_sized_view_is_dispatched_from_its_own_statistics(fr, lib, prec_name):
    import torch

    from fractal_renderer_amd import _native

    prec = 1 if prec_name == "f32" else 0
    w, h = 1920, 1080
    name = C.create_string_buffer(256)
    out = torch.empty(w * h * 3, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream()

    def render(cfg, tile=0):
        o = fr.RenderOpts(tile=tile)
        _native.check(lib.fr_render_rows_rgb8_device_opts(C.byref(cfg), prec, 0, h, out.data_ptr(), out.numel(), s.cuda_stream, C.byref(o)))
        _native.check(lib.fr_last_kernel_name(name, 256))
        torch.cuda.synchronize()
        return out.clone(), name.value

    cases = [
        # orbits of a dozen iterations everywhere -> the first pass alone (7-tile strips)
        ("thin dust", O.cli_config(w, h, O.JULIA, julia_set=(0.4, 0.4), iterations=256), 2, b"escape_first_kernel<"),
        # an interior, long orbits -> one-tile strips (what the first frame ran too)
        ("default view", O.cli_config(w, h, O.MANDELBROT, iterations=1024), 0, b"escape_strip_kernel<"),
        # short orbits but a constant the scaled loop may not use -> the strip kernel, 4-tile strips
        ("dendrite", O.cli_config(w, h, O.JULIA, julia_set=(0.0, 1.0), iterations=512), 0, b"escape_strip_kernel<"),
    ]
    _native.check(lib.fr_set_profiling(1))
    try:
        for label, ocfg, want_choice, want_kernel in cases:
            ocfg.pos.re += 1e-7 * (1 + prec)  # a view no other test of this process has rendered
            cfg = to_fr(fr, ocfg)
            want, _ = render(cfg, 8)
            assert view_state(lib, cfg, prec, h) == (0, -1, 0), label
            first, k1 = render(cfg)
            assert torch.equal(first, want), label
            assert k1.startswith(b"escape_strip_kernel") and b"1 tile" in k1, (label, k1)  # frame 1: by size
            assert view_state(lib, cfg, prec, h)[0] in (1, 3), label                       # its sample is on its way
            wait_for_totals(lib, cfg, prec, h)
            second, k2 = render(cfg)
            st, choice, strip = view_state(lib, cfg, prec, h)
            assert (st, choice) == (2, want_choice), (label, st, choice, strip)
            assert k2.startswith(want_kernel), (label, k2)
            if label == "dendrite":
                assert strip == 4 and b"4 tiles" in k2, (label, strip, k2)
            if label == "default view":
                assert strip == 1 and b"1 tile" in k2, (label, strip, k2)
            assert torch.equal(second, want), label
            # the colour map's inputs do not make a new view (src/gui.rs:183-203: exposure, colours, smooth, inside) ...
            c2 = to_fr(fr, ocfg)
            c2.exposure, c2.inside, c2.primary_color.r = 11.0, 0, 200
            assert view_state(lib, c2, prec, h)[0] == 2, label
            # ... anything that moves orbits does
            c3 = to_fr(fr, ocfg)
            c3.pos.im += 0.25
            assert view_state(lib, c3, prec, h)[0] == 0, label
    finally:
        _native.check(lib.fr_set_profiling(0))


def test_no_sample_of_either_kind_under_stream_capture(fr, lib):
    """a view sample needs a cross-stream event (non-blocking) or a host wait (blocking); neither belongs in a
    stream capture.  A render captured into a graph takes none, replays correctly, and leaves no view on record."""
    import torch

    from fractal_renderer_amd import _native

    w, h = 4096, 2048  # large enough for the BLOCKING sample outside a capture
    ocfg = O.cli_config(w, h, O.MANDELBROT, iterations=300, pos=(-0.61234, 0.0123))
    cfg = to_fr(fr, ocfg)
    out = torch.zeros(w * h * 3, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        g.capture_begin()
        _native.check(lib.fr_render_rows_rgb8_device(C.byref(cfg), 0, 0, h, out.data_ptr(), out.numel(), side.cuda_stream))
        g.capture_end()
    torch.cuda.synchronize()
    assert view_state(lib, cfg, 0, h) == (0, -1, 0)
    assert int(out.count_nonzero()) == 0  # captured, not run
    g.replay()
    torch.cuda.synchronize()
    want = fr.get_image_rows(cfg, 0, h, 0, opts=fr.RenderOpts(tile=8))
    assert np.array_equal(out.cpu().numpy().reshape(h, w, 3), want)
