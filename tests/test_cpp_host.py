"""The C++ host-side mirror of the calc API (fractal-renderer_amd/host/fractal.hpp): builds and
links against the C ABI on CPU; runs its known-answer tests on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_host_mirror")


def build_cpp():
    import __graft_entry__ as ge

    ge.build()
    pkg = os.path.join(ROOT, "fractal-renderer_amd")
    cmd = ["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(pkg, "host"), os.path.join(ROOT, "tests", "cpp", "test_host_mirror.cpp"),
           "-L" + pkg, "-lfractal_hip", "-Wl,-rpath," + pkg, "-o", EXE]
    subprocess.run(cmd, check=True)


def test_cpp_host_mirror_builds_and_fails_loudly_without_gpu():
    build_cpp()
    import fractal_renderer_amd as fr

    if fr.device_count() == 0:
        r = subprocess.run([EXE], capture_output=True, text=True)
        assert r.returncode != 0 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_cpp_host_mirror_known_answers():
    build_cpp()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "cpp host mirror ok" in r.stdout


CLI_EXE = os.path.join(ROOT, "tests", "cpp", "fractal_cli")


def build_cli():
    import __graft_entry__ as ge

    ge.build()
    pkg = os.path.join(ROOT, "fractal-renderer_amd")
    cmd = ["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(pkg, "host"), os.path.join(pkg, "cli", "fractal_cli.cpp"),
           "-L" + pkg, "-lfractal_hip", "-Wl,-rpath," + pkg, "-o", CLI_EXE]
    subprocess.run(cmd, check=True)


def test_cli_builds_and_rejects_bad_arguments():
    build_cli()
    for args, msg in [(["-a", "julia"], "--julia-real and --julia-imaginary are required"),
                      (["--bogus"], "unknown flag"), (["-a", "newton"], "invalid algorithm name"),
                      (["--primary-color", "12345"], "failed to parse hex color"), (["1", "2", "3"], "too many positional")]:
        r = subprocess.run([CLI_EXE] + args, capture_output=True, text=True)
        assert r.returncode == 2 and msg in r.stderr, (args, r.stderr)


def _read_ppm(path):
    import numpy as np

    data = open(path, "rb").read()
    parts = data.split(b"\n", 3)
    assert parts[0] == b"P6" and parts[2] == b"255"
    w, h = map(int, parts[1].split())
    return np.frombuffer(parts[3], dtype=np.uint8).reshape(h, w, 3)


@pytest.mark.gpu
def test_cli_reproduces_reference_command_lines(tmp_path):
    """examples.md command lines (small sizes) through the CLI front end vs the oracle configured the
    way the reference's get_options would (src/lib.rs:168-226)."""
    import numpy as np

    import oracle_lib as O

    build_cli()
    cases = [
        (["96", "64"], O.cli_config(96, 64)),                                                     # "Golden"
        (["-d", "96", "64"], O.cli_config(96, 64, inside=0)),                                     # "Classic"
        (["-i", "400", "96", "64"], O.cli_config(96, 64, iterations=400)),                        # "Golden fringe"
        (["-s", "2000", "-x", "-0.74364990", "-y", "0.13188204", "-i", "800", "96", "64"],
         O.cli_config(96, 64, scale=(2000.0, 2000.0), pos=(-0.74364990, 0.13188204), iterations=800)),
        (["-a", "julia", "--julia-real", "-0.8", "--julia-imaginary", "0.156", "-i", "2000", "-s", "0.6", "-e", "30",
          "100", "50"],
         O.cli_config(100, 50, O.JULIA, julia_set=(-0.8, 0.156), iterations=2000, scale=(0.6, 0.6), exposure=30.0)),
        (["-u", "--scale-x", "0.3", "--scale-y", "0.5", "--primary-color", "ff8000", "--secondary-color", "10c020",
          "-l", "100", "--stable-limit", "1.5", "64", "64"],
         # parse_hex_rgb -> RGB::new(r, g, b) stores {r, g: b, b: g} (src/lib.rs:28, calc/src/lib.rs:129-131)
         O.cli_config(64, 64, smooth=0, scale=(0.3, 0.5), primary_color=(0xFF, 0x00, 0x80),
                      secondary_color=(0x10, 0x20, 0xC0), limit=100.0, stable_limit=1.5)),
    ]
    for i, (args, ocfg) in enumerate(cases):
        out = str(tmp_path / ("img%d" % i))
        r = subprocess.run([CLI_EXE] + args + ["-o", out, "--quiet"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        img = _read_ppm(out + ".ppm")
        O.set_log2_mode(O.LOG2_SOFT)
        try:
            want = O.get_image(ocfg)
        finally:
            O.set_log2_mode(O.LOG2_LIBM)
        assert np.array_equal(img, want), args


@pytest.mark.gpu
def test_cli_fern_and_multi_device(tmp_path):
    """-a fern through the CLI (seeded: --threads 4 --seed 7) against the oracle's restatement with the same
    RNG and the library's automatic walker count; and a Mandelbrot command line over --devices 0,0,0."""
    import numpy as np

    import oracle_lib as O

    build_cli()
    out = str(tmp_path / "fern")
    r = subprocess.run([CLI_EXE, "-a", "fern", "--threads", "4", "--seed", "7", "-i", "200000", "400", "300", "-o", out, "--quiet"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    # what get_options builds for -a fern (src/lib.rs:168-226): Config::new(fern) + CLI defaults (x = -0.6, ...)
    ocfg = O.cli_config(400, 300, O.BARNSLEY_FERN, iterations=200000)
    steps = 200000 // 4
    walkers = max(1, min(65536, steps // 256))
    assert np.array_equal(_read_ppm(out + ".ppm"), O.fern_image(ocfg, 4, 7, walkers))
    out = str(tmp_path / "multi")
    r = subprocess.run([CLI_EXE, "--devices", "0,0,0", "-i", "300", "777", "333", "-o", out, "--quiet"], capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(_read_ppm(out + ".ppm"), O.get_image(O.cli_config(777, 333, iterations=300)))


C_EXE = os.path.join(ROOT, "tests", "cpp", "test_c_abi")


def build_c():
    import __graft_entry__ as ge

    ge.build()
    pkg = os.path.join(ROOT, "fractal-renderer_amd")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "test_c_abi.c"), "-L" + pkg, "-lfractal_hip", "-Wl,-rpath," + pkg,
           "-o", C_EXE]
    subprocess.run(cmd, check=True)


def test_header_is_plain_c_and_links():
    build_c()
    import fractal_renderer_amd as fr

    if fr.device_count() == 0:
        r = subprocess.run([C_EXE], capture_output=True, text=True)
        assert r.returncode == 0 and "no device" in r.stdout, (r.returncode, r.stdout, r.stderr)


@pytest.mark.gpu
def test_c_abi_known_answers_from_c():
    build_c()
    r = subprocess.run([C_EXE], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "c abi ok", (r.returncode, r.stdout, r.stderr)
