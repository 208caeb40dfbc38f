"""fractal-renderer_amd — MI355X (gfx950) drop-in for Icelk/fractal-renderer's escape-time path.

Host-side mirror of the reference's public surface for this path, over the C ABI of
include/fractal_hip.h (libfractal_hip.so; hand-written HIP kernels in csrc/):

    reference (Rust)                                   here
    ------------------------------------------------   ---------------------------------
    calc::Config, Config::new(algo)  lib.rs:21-69      Config, Config.new(algo)
    calc::Algo                       lib.rs:150-154    Algo
    calc::Imaginary / calc::RGB      lib.rs:79-82,121  Imaginary / RGB (RGB.new(r, b, g))
    calc::recursive                  lib.rs:245-257    recursive(iterations, start, c, limit)
    calc::get_recursive_pixel        lib.rs:199-235    get_recursive_pixel(config, x, y)
    get_image (Mandelbrot|Julia arm) src/lib.rs:253    get_image(config)

There is no CPU fallback anywhere in this package: without the built library and a HIP device
every compute call raises.
"""
import ctypes as C
import enum

import numpy as np

from . import _native
from ._native import RGB, FractalHipError, Imaginary  # noqa: F401

__all__ = [
    "Algo", "Config", "Imaginary", "RGB", "Precision", "FractalHipError",
    "get_image", "get_image_rows", "get_image_rgba", "get_recursive_pixel", "recursive", "recursive_batch",
    "escape_rows", "colour_image", "count_iterations", "init", "shutdown", "device_count", "device_name",
    "RenderOpts", "init_devices", "get_image_multi", "multi_stats", "build_id", "get_image_fern",
]


class Algo(enum.IntEnum):
    """calc/src/lib.rs:150-154; from_str mirrors FromStr (:165-179)."""

    Mandelbrot = 0
    BarnsleyFern = 1
    Julia = 2

    @classmethod
    def from_str(cls, s):
        t = s.lower()
        if t == "mandelbrot":
            return cls.Mandelbrot
        if t in ("fern", "barnsleyfern"):
            return cls.BarnsleyFern
        if t == "julia":
            return cls.Julia
        raise ValueError("invalid algorithm name")


class Precision(enum.IntEnum):
    F64 = 0  # the reference's arithmetic
    F32 = 1  # build-defined fast path (include/fractal_hip.h, fr_precision)


class Config(_native.fr_config):
    """calc::Config (calc/src/lib.rs:21-37).  Fields keep the reference's names; colours hold the
    stored RGB struct fields."""

    @classmethod
    def new(cls, algo=Algo.Mandelbrot):
        """Config::new(algo) — calc/src/lib.rs:39-69 (filled in by the library)."""
        cfg = cls()
        _native.load().fr_config_new(C.byref(cfg), int(algo))
        return cfg

    def clone(self):
        other = type(self)()
        C.memmove(C.byref(other), C.byref(self), C.sizeof(self))
        return other


class RenderOpts(_native.fr_render_opts):
    """fr_render_opts: implementation selectors of ONE call (none changes an output byte).
    RenderOpts(tile=9, cycle_shortcut=1) starts from the process defaults."""

    def __init__(self, **kw):
        super().__init__()
        _native.load().fr_render_opts_init(C.byref(self))
        for k, v in kw.items():
            if k not in dict(self._fields_) or k == "size":
                raise AttributeError(k)
            setattr(self, k, v)


def build_id():
    return _native.load().fr_build_id().decode()


def init_devices(devices):
    """fr_init_devices: the device set of the multi-GPU get_image; an index may repeat (logical devices)."""
    arr = (C.c_int * len(devices))(*devices)
    _native.check(_native.load().fr_init_devices(arr, len(devices)))


def get_image_multi(config, precision=0, block_rows=0, out=None):
    """get_image across the device set into a host array: every device DMAs its row blocks to their
    final place (fr_render_rgb8_multi)."""
    if out is None:
        out = np.empty((config.height, config.width, 3), dtype=np.uint8)
    _native.check(_native.load().fr_render_rgb8_multi(C.byref(config), int(precision), block_rows, out.ctypes.data,
                                                      out.nbytes))
    return out


def multi_stats():
    st = _native.fr_multi_stats()
    _native.check(_native.load().fr_multi_last_stats(C.byref(st)))
    n = st.n_devices
    return {"n_devices": n, "kernels": list(st.kernels[:n]), "kernel_ms": list(st.kernel_ms[:n]),
            "rows": list(st.rows[:n]), "wall_ms": st.wall_ms, "transfer_span_ms": list(st.transfer_span_ms[:n]),
            "job_ms": list(st.job_ms[:n]), "bytes_moved": list(st.bytes_moved[:n])}


def init(device=-1):
    _native.check(_native.load().fr_init(device))


def shutdown():
    _native.check(_native.load().fr_shutdown())


def device_count():
    n = C.c_int(0)
    _native.check(_native.load().fr_device_count(C.byref(n)))
    return n.value


def device_name():
    buf = C.create_string_buffer(256)
    _native.check(_native.load().fr_device_name(buf, len(buf)))
    return buf.value.decode()


def get_image_rows(config, y0, y1, precision=Precision.F64, out=None, opts=None):
    """Rows [y0, y1) of get_image — the unit of the reference's rayon loop (src/lib.rs:256-264).
    Returns uint8 [y1-y0, width, 3]."""
    if out is None:
        out = np.empty((max(int(y1) - int(y0), 0), config.width, 3), dtype=np.uint8)
    _native.check(
        _native.load().fr_render_rows_rgb8_opts(C.byref(config), int(precision), y0, y1, out.ctypes.data, out.nbytes,
                                                C.byref(opts) if opts is not None else None)
    )
    return out


def get_image(config, precision=Precision.F64):
    """get_image(&Config) -> Vec<RGB> (src/lib.rs:253-270): uint8 [height, width, 3], row-major,
    bytes r,g,b.  Algo.BarnsleyFern is outside this path (random IFS, src/lib.rs:271-319): the
    per-pixel function returns BLACK for it (calc/src/lib.rs:211) and so does this."""
    out = np.empty((config.height, config.width, 3), dtype=np.uint8)
    if int(precision) == Precision.F64:
        _native.check(_native.load().fr_render_rgb8(C.byref(config), out.ctypes.data, out.nbytes))
        return out
    return get_image_rows(config, 0, config.height, precision, out)


def get_image_fern(config, threads=1, seed=0, walkers=0):
    """get_image's Algo::BarnsleyFern arm (src/lib.rs:271-319, 417-463) on the GPU: uint8 [height, width, 3].
    threads = the rayon thread count being modelled (the reference returns ONE thread's image of
    iterations / threads points); seed keys the build's deterministic RNG; walkers = parallel orbits (0 = auto)."""
    out = np.empty((config.height, config.width, 3), dtype=np.uint8)
    _native.check(_native.load().fr_render_fern_rgb8(C.byref(config), threads, seed, walkers, out.ctypes.data, out.nbytes))
    return out


def get_image_rgba(config, precision=Precision.F64, out=None):
    """get_image as RGBA8 (alpha 255): uint8 [height, width, 4] — the GUI's upload format
    (src/gui.rs:71-72) produced on the device."""
    if out is None:
        out = np.empty((config.height, config.width, 4), dtype=np.uint8)
    _native.check(
        _native.load().fr_render_rows_rgba8(C.byref(config), int(precision), 0, config.height, out.ctypes.data,
                                            out.nbytes)
    )
    return out


def get_recursive_pixel(config, x, y, precision=Precision.F64):
    """calc::get_recursive_pixel(&Config, x, y) -> RGB (calc/src/lib.rs:199-235)."""
    out = RGB()
    _native.check(_native.load().fr_pixel_p(C.byref(config), int(precision), x, y, C.byref(out)))
    return out


def recursive(iterations, start, c, limit):
    """calc::recursive(iterations, start, c, limit) -> (Imaginary, u32) (calc/src/lib.rs:245-257)."""
    pos, it = Imaginary(), C.c_uint32(0)
    _native.check(
        _native.load().fr_recursive(iterations, Imaginary(*start), Imaginary(*c), limit, C.byref(pos), C.byref(it))
    )
    return pos, it.value


def recursive_batch(iterations, start, c, limit, precision=Precision.F64):
    """recursive() over arrays: start, c float64 [n, 2] -> (pos float64 [n, 2], iters uint32 [n])."""
    start = np.ascontiguousarray(start, dtype=np.float64).reshape(-1, 2)
    c = np.ascontiguousarray(c, dtype=np.float64).reshape(-1, 2)
    if start.shape != c.shape:
        raise ValueError("start and c must have the same shape")
    n = start.shape[0]
    pos = np.empty((n, 2), dtype=np.float64)
    it = np.empty(n, dtype=np.uint32)
    _native.check(
        _native.load().fr_recursive_batch(iterations, start.ctypes.data, c.ctypes.data, n, limit, int(precision),
                                          pos.ctypes.data, it.ctypes.data)
    )
    return pos, it


def escape_rows(config, y0=0, y1=None, precision=Precision.F64):
    """recursive() results of every pixel of rows [y0, y1): (z float64 [rows, width, 2],
    iters uint32 [rows, width])."""
    y1 = config.height if y1 is None else y1
    z = np.empty((y1 - y0, config.width, 2), dtype=np.float64)
    it = np.empty((y1 - y0, config.width), dtype=np.uint32)
    _native.check(
        _native.load().fr_escape_rows(C.byref(config), int(precision), y0, y1, z.ctypes.data, it.ctypes.data)
    )
    return z, it


def colour_image(config, z, iters):
    """The colour map alone (calc/src/lib.rs:214-234) over stored recursive() results: z float64
    [..., 2], iters uint32 [...] -> uint8 [..., 3].  Re-colouring (exposure, colours, smooth, inside)
    without re-iterating."""
    z = np.ascontiguousarray(z, dtype=np.float64)
    iters = np.ascontiguousarray(iters, dtype=np.uint32)
    if z.shape[:-1] != iters.shape or z.shape[-1] != 2:
        raise ValueError("z must be [..., 2] and iters [...]")
    out = np.empty(iters.shape + (3,), dtype=np.uint8)
    _native.check(
        _native.load().fr_colour_rgb8(C.byref(config), z.ctypes.data, iters.ctypes.data, iters.size, out.ctypes.data,
                                      out.nbytes)
    )
    return out


def count_iterations(config, y0=0, y1=None, sx=1, sy=1, precision=Precision.F64):
    """Exact Σ executed iterations over the sampled pixels; returns (total, pixels)."""
    y1 = config.height if y1 is None else y1
    total, npx = C.c_uint64(0), C.c_uint64(0)
    _native.check(
        _native.load().fr_count_iterations(C.byref(config), int(precision), y0, y1, sx, sy, C.byref(total),
                                           C.byref(npx))
    )
    return total.value, npx.value
