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
