"""CPU-side checks of the C-ABI boundary (no compute calls): the library loads, exports every
symbol include/fractal_hip.h declares, mirrors calc::Config, and fails loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fr():
    import __graft_entry__ as ge

    ge.build()
    import fractal_renderer_amd

    return fractal_renderer_amd


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "fractal_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(fr):
    from fractal_renderer_amd import _native

    lib = _native.load()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "libfractal_hip.so does not export %s" % s
    assert set(syms) == set(_native.PROTOTYPES), "python prototypes out of sync with the header"
    assert lib.fr_abi_version() == 3
    assert len(lib.fr_build_id()) == 16


def test_config_layout_matches_reference_fields(fr):
    from fractal_renderer_amd import _native

    assert C.sizeof(_native.fr_config) == 104 == C.sizeof(O.Config)
    names = [f[0] for f in _native.fr_config._fields_]
    # calc/src/lib.rs:21-37, declaration order
    assert names == ["algo", "width", "height", "iterations", "limit", "stable_limit", "pos", "scale", "exposure",
                     "inside", "smooth", "primary_color", "secondary_color", "color_weight", "julia_set"]
    for name in names:
        assert getattr(_native.fr_config, name).offset == getattr(O.Config, name).offset


@pytest.mark.parametrize("algo", [0, 1, 2])
def test_config_new_matches_oracle(fr, algo):
    a = fr.Config.new(algo)
    b = O.config_new(algo)
    assert bytes(a) == bytes(b)


def test_rgb_new_argument_order(fr):
    # RGB::new(r, b, g) — calc/src/lib.rs:129-131
    assert tuple(fr.RGB.new(40, 40, 255)) == (40, 255, 40)
    assert tuple(fr.Config.new().primary_color) == (40, 255, 40)
    assert tuple(fr.Config.new().secondary_color) == (240, 0, 170)


def test_algo_from_str(fr):
    # calc/src/lib.rs:165-179
    assert fr.Algo.from_str("Mandelbrot") == fr.Algo.Mandelbrot
    assert fr.Algo.from_str("FERN") == fr.Algo.BarnsleyFern == fr.Algo.from_str("barnsleyfern")
    assert fr.Algo.from_str("julia") == fr.Algo.Julia
    with pytest.raises(ValueError):
        fr.Algo.from_str("newton")


def test_block_cyclic_row_count(fr):
    from fractal_renderer_amd import _native

    f = _native.load().fr_block_cyclic_rows
    assert f(100, 8, 0, 1) == 100
    assert sum(f(100, 8, r, 3) for r in range(3)) == 100
    assert f(100, 8, 0, 3) == 8 * 4 + 4  # blocks 0,3,6,9,12(4 rows: 96..99)
    assert f(100, 8, 1, 3) == 8 * 4  # blocks 1,4,7,10
    assert f(0, 8, 0, 1) == 0 and f(10, 0, 0, 1) == 0 and f(10, 4, 5, 2) == 0
    assert sum(f(65536, 64, r, 8) for r in range(8)) == 65536


def test_argument_errors_need_no_device(fr):
    from fractal_renderer_amd import _native

    lib = _native.load()
    cfg = fr.Config.new()
    buf = np.zeros(16, dtype=np.uint8)
    assert lib.fr_render_rows_rgb8(C.byref(cfg), 0, 5, 4, buf.ctypes.data, buf.nbytes) == _native.FR_ERR_INVALID_ARGUMENT
    assert b"y0 > y1" in lib.fr_last_error()
    assert lib.fr_render_rows_rgb8(C.byref(cfg), 0, 0, cfg.height + 1, buf.ctypes.data, buf.nbytes) == _native.FR_ERR_INVALID_ARGUMENT
    assert lib.fr_render_rows_rgb8(C.byref(cfg), 7, 0, 1, buf.ctypes.data, buf.nbytes) == _native.FR_ERR_INVALID_ARGUMENT
    assert lib.fr_render_rows_rgb8(C.byref(cfg), 0, 0, 1, buf.ctypes.data, buf.nbytes) == _native.FR_ERR_BUFFER_TOO_SMALL
    assert lib.fr_render_rgb8(None, buf.ctypes.data, buf.nbytes) == _native.FR_ERR_INVALID_ARGUMENT
    # y0 == y1 renders nothing and is legal even without a device (src/lib.rs:256: empty range)
    assert lib.fr_render_rows_rgb8(C.byref(cfg), 0, 3, 3, None, 0) == _native.FR_OK
    assert lib.fr_set_tile(1234) == _native.FR_ERR_INVALID_ARGUMENT
    assert lib.fr_set_tile(0) == _native.FR_OK


def test_no_cpu_fallback_without_device(fr):
    """On a box without a GPU every compute entry point must fail loudly, not compute on the CPU."""
    if fr.device_count() > 0:
        pytest.skip("a HIP device is present")
    cfg = fr.Config.new()
    cfg.width, cfg.height = 8, 8
    with pytest.raises(fr.FractalHipError) as e:
        fr.get_image(cfg)
    assert e.value.code == 3  # FR_ERR_NO_DEVICE
    with pytest.raises(fr.FractalHipError):
        fr.get_recursive_pixel(cfg, 0, 0)
    with pytest.raises(fr.FractalHipError):
        fr.recursive(50, (2, 0), (2, 0), 65536)
    with pytest.raises(fr.FractalHipError):
        fr.count_iterations(cfg)


def test_product_does_not_reference_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "fractal-renderer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", ".inc")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "fractal_oracle" not in text and "oracle_lib" not in text and "fro_" not in text, f


def test_rccl_load_failure_is_an_error_code_not_a_crash():
    """Rccl::load() built its message from two dlerror() calls (the second returns NULL:
    std::string + NULL crashed exactly when librccl could not be loaded).  Forced here with FR_RCCL_LIBRARY
    in a child process (the variable is read when the load is attempted; a crash would take pytest down)."""
    import subprocess
    import sys

    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import fractal_renderer_amd as fr\n"
        "from fractal_renderer_amd import _native\n"
        "lib = _native.load()\n"
        "rc = lib.fr_debug_rccl_probe()\n"
        "print('rc', rc, lib.fr_last_error().decode())\n" % ROOT
    )
    env = dict(os.environ, FR_RCCL_LIBRARY="/nonexistent/librccl-not-here.so")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr)
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("rc ")][0]
    assert int(line.split()[1]) == 4, line  # FR_ERR_HIP
    assert "cannot load librccl" in line and "librccl-not-here" in line, line


def test_bench_supervisor_kills_a_stalled_child_and_falls_back():
    """VERDICT r02 #2b: plain `python3 bench.py --gpus N` runs the measurement in a fresh child process with a time
    limit; a child that hangs is killed (whole process group) and a second fresh child tries the peer-DMA gather.
    Here the first child is made to stall (FR_BENCH_TEST_STALL=rccl) and ITS limit is 3 s (a healthy child keeps a
    long one: its first `import torch` on a cold box can take a minute); the fallback child then
    reports, on this GPU-less box, that there are no devices — through the supervisor, with the fallback recorded."""
    import json
    import subprocess
    import sys
    import time

    env = dict(os.environ, FR_BENCH_TEST_STALL="rccl", FR_BENCH_TEST_STALL_TIMEOUT="3", FR_BENCH_CHILD_TIMEOUT="240")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    took = time.time() - t0
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert p.returncode == 0, (p.returncode, p.stdout[-500:], p.stderr[-500:])
    assert took < 280, took
    assert d["n_gpus"] == 2 and d["value"] is None and "needs 2 devices" in d["error"]
    fb = d["fallback"]
    assert "hung and was killed" in fb["reason"] and fb["attempts"][0]["timed_out"] is True
    assert fb["attempts"][1]["what"].startswith("peer-to-peer") and fb["attempts"][1]["exit_code"] == 0
    # both children stalled: a parsable error line and a non-zero exit code, never a hang
    env["FR_BENCH_TEST_STALL"] = "peer"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--gather", "peer"], env=env,
                       capture_output=True, text=True, timeout=300)
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert p.returncode == 1 and d["value"] is None and d["error"] == "every measured child process failed"
