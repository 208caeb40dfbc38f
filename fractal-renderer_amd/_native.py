"""ctypes loader of libfractal_hip.so (the C ABI of include/fractal_hip.h).

There is no fallback: a missing or unloadable library raises, and every compute call raises
FractalHipError when the HIP path fails (no GPU, HIP error).
"""
import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libfractal_hip.so")

FR_OK, FR_ERR_INVALID_ARGUMENT, FR_ERR_BUFFER_TOO_SMALL, FR_ERR_NO_DEVICE, FR_ERR_HIP = 0, 1, 2, 3, 4
STATUS_NAMES = {
    1: "FR_ERR_INVALID_ARGUMENT",
    2: "FR_ERR_BUFFER_TOO_SMALL",
    3: "FR_ERR_NO_DEVICE",
    4: "FR_ERR_HIP",
    5: "FR_ERR_UNSUPPORTED_ALGO",
}


class FractalHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("%s: %s" % (STATUS_NAMES.get(code, "fr_status %d" % code), message))
        self.code = code


class Imaginary(C.Structure):
    """calc/src/lib.rs:79-82"""

    _fields_ = [("re", C.c_double), ("im", C.c_double)]

    def __iter__(self):
        return iter((self.re, self.im))


class RGB(C.Structure):
    """calc/src/lib.rs:121-125 — the stored fields (see RGB.new for the constructor quirk)."""

    _fields_ = [("r", C.c_uint8), ("g", C.c_uint8), ("b", C.c_uint8)]

    @classmethod
    def new(cls, r, b, g):
        """RGB::new(r, b, g) — calc/src/lib.rs:129-131: the second parameter is BLUE."""
        return cls(r, g, b)

    def __iter__(self):
        return iter((self.r, self.g, self.b))

    def __eq__(self, other):
        return tuple(self) == tuple(other)

    def __repr__(self):
        return "RGB { r: %d, g: %d, b: %d }" % tuple(self)


class fr_config(C.Structure):
    """include/fractal_hip.h fr_config == calc::Config (calc/src/lib.rs:21-37)."""

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


class fr_render_opts(C.Structure):
    """include/fractal_hip.h fr_render_opts: the implementation selectors of ONE call."""

    _fields_ = [
        ("size", C.c_uint32),
        ("tile", C.c_int32),
        ("loop_mode", C.c_int32),
        ("palette", C.c_int32),
        ("cycle_shortcut", C.c_int32),
        ("refill_minrun", C.c_int32),
        ("refill_quit16", C.c_int32),
        ("colour_filter", C.c_int32),
    ]


FR_MAX_DEVICES = 16
FR_GATHER_PEER_COPY, FR_GATHER_RCCL = 0, 1


class fr_multi_stats(C.Structure):
    _fields_ = [
        ("n_devices", C.c_uint32),
        ("kernels", C.c_uint32 * FR_MAX_DEVICES),
        ("kernel_ms", C.c_float * FR_MAX_DEVICES),
        ("rows", C.c_uint64 * FR_MAX_DEVICES),
        ("wall_ms", C.c_double),
        ("transfer_span_ms", C.c_float * FR_MAX_DEVICES),
        ("job_ms", C.c_double * FR_MAX_DEVICES),
        ("bytes_moved", C.c_uint64 * FR_MAX_DEVICES),
    ]


_OPTS = C.POINTER(fr_render_opts)

# name -> (restype, argtypes); every symbol include/fractal_hip.h declares
PROTOTYPES = {
    "fr_abi_version": (C.c_int, []),
    "fr_build_id": (C.c_char_p, []),
    "fr_render_opts_init": (None, [_OPTS]),
    "fr_render_rows_rgb8_device_opts": (
        C.c_int,
        [C.POINTER(fr_config), C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, _OPTS],
    ),
    "fr_render_rows_rgba8_device_opts": (
        C.c_int,
        [C.POINTER(fr_config), C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, _OPTS],
    ),
    "fr_render_rows_rgb8_opts": (
        C.c_int, [C.POINTER(fr_config), C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, _OPTS]),
    "fr_render_block_cyclic_range_rgb8_device_opts": (
        C.c_int,
        [C.POINTER(fr_config), C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_size_t,
         C.c_void_p, C.POINTER(C.c_uint64), _OPTS],
    ),
    "fr_render_fern_rgb8": (C.c_int, [C.POINTER(fr_config), C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p, C.c_size_t]),
    "fr_init_devices": (C.c_int, [C.POINTER(C.c_int), C.c_int]),
    "fr_multi_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "fr_set_multi_root_share": (C.c_int, [C.c_int]),
    "fr_render_rgb8_multi": (C.c_int, [C.POINTER(fr_config), C.c_int, C.c_uint32, C.c_void_p, C.c_size_t]),
    "fr_render_rgb8_multi_device": (
        C.c_int, [C.POINTER(fr_config), C.c_int, C.c_uint32, C.c_int, C.c_void_p, C.c_size_t]),
    "fr_multi_last_stats": (C.c_int, [C.POINTER(fr_multi_stats)]),
    "fr_debug_rccl_selftest": (C.c_int, [C.c_size_t]),
    "fr_debug_rccl_probe": (C.c_int, []),
    "fr_set_dispatch_sampling": (C.c_int, [C.c_int]),
    "fr_debug_sample_view": (C.c_int, [C.POINTER(fr_config), C.c_int, C.POINTER(C.c_double)]),
    "fr_debug_view_choice": (C.c_int, [C.POINTER(fr_config), C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_int),
                                       C.POINTER(C.c_int), C.POINTER(C.c_uint32)]),
    "fr_debug_inject_multi_failure": (C.c_int, [C.c_int, C.c_int]),
    "fr_pin_host_buffer": (C.c_int, [C.c_void_p, C.c_size_t]),
    "fr_unpin_host_buffer": (C.c_int, [C.c_void_p]),
    "fr_last_kernel_name": (C.c_int, [C.c_char_p, C.c_size_t]),
    "fr_set_colour_filter": (C.c_int, [C.c_int]),
    "fr_init": (C.c_int, [C.c_int]),
    "fr_shutdown": (C.c_int, []),
    "fr_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "fr_device_name": (C.c_int, [C.c_char_p, C.c_size_t]),
    "fr_last_error": (C.c_char_p, []),
    "fr_config_new": (None, [C.POINTER(fr_config), C.c_uint32]),
    "fr_render_rgb8": (C.c_int, [C.POINTER(fr_config), C.c_void_p, C.c_size_t]),
    "fr_render_rows_rgb8": (C.c_int, [C.POINTER(fr_config), C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]),
    "fr_render_rows_rgb8_device": (
        C.c_int,
        [C.POINTER(fr_config), C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "fr_render_rows_rgba8": (C.c_int, [C.POINTER(fr_config), C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]),
    "fr_render_rows_rgba8_device": (
        C.c_int,
        [C.POINTER(fr_config), C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "fr_render_block_cyclic_rgb8_device": (
        C.c_int,
        [C.POINTER(fr_config), C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p,
         C.POINTER(C.c_uint64)],
    ),
    "fr_render_block_cyclic_range_rgb8_device": (
        C.c_int,
        [C.POINTER(fr_config), C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_size_t,
         C.c_void_p, C.POINTER(C.c_uint64)],
    ),
    "fr_render_block_cyclic_rgb8": (
        C.c_int,
        [C.POINTER(fr_config), C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t,
         C.POINTER(C.c_uint64)],
    ),
    "fr_block_cyclic_rows": (C.c_uint64, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "fr_pixel": (C.c_int, [C.POINTER(fr_config), C.c_uint32, C.c_uint32, C.POINTER(RGB)]),
    "fr_pixel_p": (C.c_int, [C.POINTER(fr_config), C.c_int, C.c_uint32, C.c_uint32, C.POINTER(RGB)]),
    "fr_recursive": (
        C.c_int,
        [C.c_uint32, Imaginary, Imaginary, C.c_double, C.POINTER(Imaginary), C.POINTER(C.c_uint32)],
    ),
    "fr_recursive_batch": (
        C.c_int,
        [C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_int, C.c_void_p, C.c_void_p],
    ),
    "fr_escape_rows": (C.c_int, [C.POINTER(fr_config), C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "fr_colour_rgb8": (C.c_int, [C.POINTER(fr_config), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "fr_colour_rgb8_device": (
        C.c_int, [C.POINTER(fr_config), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "fr_count_iterations": (
        C.c_int,
        [C.POINTER(fr_config), C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64),
         C.POINTER(C.c_uint64)],
    ),
    "fr_set_profiling": (C.c_int, [C.c_int]),
    "fr_last_kernel_ms": (C.c_int, [C.POINTER(C.c_float)]),
    "fr_set_tile": (C.c_int, [C.c_int]),
    "fr_set_loop_mode": (C.c_int, [C.c_int]),
    "fr_debug_loop_plan": (C.c_int, [C.POINTER(fr_config), C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_double),
                                     C.POINTER(C.c_uint32)]),
    "fr_set_palette": (C.c_int, [C.c_int]),
    "fr_set_cycle_shortcut": (C.c_int, [C.c_int]),
    "fr_set_refill_policy": (C.c_int, [C.c_int, C.c_int]),
    "fr_debug_math": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]),
    "fr_debug_set_queue_trace": (C.c_int, [C.c_void_p]),
    "fr_debug_set_two_pass_capacity": (C.c_int, [C.c_uint32]),
}

_lib = None


def load():
    """Load libfractal_hip.so and bind every prototype.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing — build it first (python -c 'import __graft_entry__ as g; g.build()'); "
            "this package has no CPU fallback" % LIB_PATH
        )
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc):
    if rc != FR_OK:
        msg = load().fr_last_error()
        raise FractalHipError(rc, msg.decode() if msg else "")
