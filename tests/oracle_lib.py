"""ctypes binding of the CPU oracle (oracle/libfractal_oracle.so) — test infrastructure only.

The oracle restates calc/src/lib.rs:83-257 and src/lib.rs:253-270 in C (see
oracle/fractal_oracle.h).  Nothing in the product imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libfractal_oracle.so")

MANDELBROT, BARNSLEY_FERN, JULIA = 0, 1, 2
F64, F32 = 0, 1
LOG2_LIBM, LOG2_SOFT = 0, 1


class Imaginary(C.Structure):
    _fields_ = [("re", C.c_double), ("im", C.c_double)]


class RGB(C.Structure):
    _fields_ = [("r", C.c_uint8), ("g", C.c_uint8), ("b", C.c_uint8)]

    def bytes(self):
        return (self.r, self.g, self.b)


class Config(C.Structure):
    """calc/src/lib.rs:21-37, field for field (same layout as fr_config)."""

    _fields_ = [
        ("algo", C.c_uint32),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("iterations", C.c_uint32),
        ("limit", C.c_double),
        ("stable_limit", C.c_double),
        ("pos", Imaginary),
        ("scale", Imaginary),
        ("exposure", C.c_double),
        ("inside", C.c_uint8),
        ("smooth", C.c_uint8),
        ("primary_color", RGB),
        ("secondary_color", RGB),
        ("color_weight", C.c_double),
        ("julia_set", Imaginary),
    ]


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(
        os.path.join(ORACLE_DIR, "fractal_oracle.c")
    ):
        build_oracle()
    L = C.CDLL(ORACLE_SO)
    L.fro_rgb_new.restype = RGB
    L.fro_rgb_new.argtypes = [C.c_uint8, C.c_uint8, C.c_uint8]
    L.fro_config_new.argtypes = [C.POINTER(Config), C.c_uint32]
    L.fro_recursive.restype = C.c_uint32
    L.fro_recursive.argtypes = [C.c_uint32, Imaginary, Imaginary, C.c_double, C.POINTER(Imaginary)]
    L.fro_recursive_f32.restype = C.c_uint32
    L.fro_recursive_f32.argtypes = L.fro_recursive.argtypes
    L.fro_xy_to_imaginary.restype = Imaginary
    L.fro_xy_to_imaginary.argtypes = [C.POINTER(Config), C.c_uint32, C.c_uint32]
    L.fro_get_recursive_pixel.restype = RGB
    L.fro_get_recursive_pixel.argtypes = [C.POINTER(Config), C.c_uint32, C.c_uint32]
    L.fro_get_recursive_pixel_p.restype = RGB
    L.fro_get_recursive_pixel_p.argtypes = [C.POINTER(Config), C.c_int, C.c_uint32, C.c_uint32]
    L.fro_get_image_rows.restype = C.c_int
    L.fro_get_image_rows.argtypes = [C.POINTER(Config), C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int]
    L.fro_escape_rows.restype = C.c_int
    L.fro_escape_rows.argtypes = [C.POINTER(Config), C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
    L.fro_colour_rows.restype = C.c_int
    L.fro_colour_rows.argtypes = [C.POINTER(Config), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
    L.fro_sample_image.restype = C.c_uint64
    L.fro_sample_image.argtypes = [C.POINTER(Config), C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int,
                                   C.POINTER(C.c_uint64)]
    L.fro_count_iterations_rows.restype = C.c_uint64
    L.fro_count_iterations_rows.argtypes = [C.POINTER(Config), C.c_int, C.c_uint32, C.c_uint32, C.c_int]
    L.fro_set_log2_mode.argtypes = [C.c_int]
    L.fro_get_log2_mode.restype = C.c_int
    L.fro_log2.restype = C.c_double
    L.fro_log2.argtypes = [C.c_double]
    L.fro_fern_image.restype = C.c_int
    L.fro_fern_image.argtypes = [C.POINTER(Config), C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p]
    _lib = L
    return L


def config_new(algo=MANDELBROT, **kw):
    """Config::new(algo) (calc/src/lib.rs:39-69) with keyword overrides."""
    cfg = Config()
    lib().fro_config_new(C.byref(cfg), algo)
    apply_overrides(cfg, kw)
    return cfg


def apply_overrides(cfg, kw):
    for k, v in kw.items():
        if k in ("pos", "scale", "julia_set"):
            setattr(cfg, k, Imaginary(*v))
        elif k in ("primary_color", "secondary_color"):
            # given as the STORED struct fields (r, g, b)
            setattr(cfg, k, RGB(*v))
        else:
            if not hasattr(cfg, k):
                raise AttributeError(k)
            setattr(cfg, k, v)
    return cfg


def cli_config(width=750, height=500, algo=MANDELBROT, **kw):
    """The Config the reference CLI builds when a flag is not given (src/lib.rs:34-226):
    limit 65536, stable_limit 2, pos (-0.6, 0) (0 for julia), scale 0.4, exposure 5."""
    cfg = config_new(algo)
    cfg.width, cfg.height = width, height
    cfg.exposure = 5.0
    cfg.pos = Imaginary(0.0 if algo == JULIA else -0.6, 0.0)
    apply_overrides(cfg, kw)
    return cfg


def recursive(iterations, start, c, limit, f32=False):
    out = Imaginary()
    fn = lib().fro_recursive_f32 if f32 else lib().fro_recursive
    it = fn(iterations, Imaginary(*start), Imaginary(*c), limit, C.byref(out))
    return (out.re, out.im), it


def get_recursive_pixel(cfg, x, y, precision=F64):
    return lib().fro_get_recursive_pixel_p(C.byref(cfg), precision, x, y).bytes()


def get_image(cfg, precision=F64, y0=0, y1=None, threads=0):
    """get_image (src/lib.rs:253-270) -> uint8 array [rows, width, 3]."""
    y1 = cfg.height if y1 is None else y1
    out = np.empty((y1 - y0, cfg.width, 3), dtype=np.uint8)
    lib().fro_get_image_rows(C.byref(cfg), precision, y0, y1, out.ctypes.data, threads)
    return out


def escape_rows(cfg, precision=F64, y0=0, y1=None, threads=0):
    y1 = cfg.height if y1 is None else y1
    z = np.empty((y1 - y0, cfg.width, 2), dtype=np.float64)
    it = np.empty((y1 - y0, cfg.width), dtype=np.uint32)
    lib().fro_escape_rows(C.byref(cfg), precision, y0, y1, z.ctypes.data, it.ctypes.data, threads)
    return z, it


def colour_rows(cfg, z, it, threads=0):
    """The colour map alone (calc/src/lib.rs:214-234) over stored recursive() results -> uint8 [..., 3]."""
    z = np.ascontiguousarray(z, dtype=np.float64)
    it = np.ascontiguousarray(it, dtype=np.uint32)
    out = np.empty(it.shape + (3,), dtype=np.uint8)
    lib().fro_colour_rows(C.byref(cfg), z.ctypes.data, it.ctypes.data, it.size, out.ctypes.data, threads)
    return out


def sample_image(cfg, sx, sy, precision=F64, threads=0, colours=True):
    ncols = (cfg.width + sx - 1) // sx
    nrows = (cfg.height + sy - 1) // sy
    out = np.empty((nrows, ncols, 3), dtype=np.uint8) if colours else None
    npx = C.c_uint64(0)
    total = lib().fro_sample_image(C.byref(cfg), precision, sx, sy, out.ctypes.data if colours else None, threads,
                                   C.byref(npx))
    return int(total), int(npx.value), out


def count_iterations(cfg, precision=F64, y0=0, y1=None, threads=0):
    y1 = cfg.height if y1 is None else y1
    return int(lib().fro_count_iterations_rows(C.byref(cfg), precision, y0, y1, threads))


def set_log2_mode(mode):
    lib().fro_set_log2_mode(mode)


def log2(x):
    return lib().fro_log2(x)


def fern_image(cfg, threads=1, seed=0, walkers=1):
    """get_image's BarnsleyFern arm (src/lib.rs:271-319, 417-463) with the deterministic RNG; walkers == 1 is
    the reference's single sequential orbit."""
    out = np.empty((cfg.height, cfg.width, 3), dtype=np.uint8)
    if lib().fro_fern_image(C.byref(cfg), threads, seed, walkers, out.ctypes.data) != 0:
        raise ValueError("degenerate fern arguments")
    return out
