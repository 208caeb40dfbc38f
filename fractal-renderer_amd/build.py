"""Build recipe of libfractal_hip.so (hipcc, gfx950 only), in-tree so the .so travels with the repo.

-ffp-contract=off is REQUIRED for bit parity with the reference (Rust never fuses a*b+c); see
csrc/fr_kernels.hip.
"""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libfractal_hip.so")
SOURCES = ["fr_kernels.hip", "fr_api.hip"]
DEPS = SOURCES + ["fr_kernels.h", "fr_math.h", "fr_log2_table.inc", os.path.join("..", "..", "include", "fractal_hip.h")]
HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-ffp-contract=off",
    "-fno-fast-math",
    "-fPIC",
    "-shared",
    "-Wall",
    "-Wextra",
]


def find_hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS) or os.path.getmtime(__file__) > t


def build_extension(force=False, verbose=False):
    """Compile csrc/*.hip into libfractal_hip.so.  Returns the path."""
    if not force and not is_stale():
        return LIB_PATH
    cmd = [find_hipcc()] + HIPCC_FLAGS + ["-o", LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB_PATH


if __name__ == "__main__":
    print(build_extension(force=True, verbose=True))
