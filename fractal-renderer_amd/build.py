"""Build recipe of libfractal_hip.so (hipcc, gfx950 only), in-tree so the .so travels with the repo.

-ffp-contract=off is REQUIRED for bit parity with the reference (Rust never fuses a*b+c); see
csrc/fr_kernels.hip.

Staleness is decided by CONTENT: the SHA-256 of every source, header and flag is compiled into the
library (fr_build_id()) and kept beside it in libfractal_hip.so.id, so "which sources was the .so that
ran the tests built from" has an unambiguous answer (bench.py prints it).
"""
import hashlib
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libfractal_hip.so")
ID_PATH = LIB_PATH + ".id"
SOURCES = ["fr_kernels.hip", "fr_api.hip", "fr_host.hip", "fr_multi.hip", "fr_fern.hip"]
DEPS = SOURCES + ["fr_kernels.h", "fr_ctx.h", "fr_math.h", "fr_log2_table.inc",
                  os.path.join("..", "..", "include", "fractal_hip.h")]
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
    "-pthread",
]
LINK_FLAGS = ["-ldl"]


def find_hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def source_id():
    """SHA-256 (first 16 hex digits) over the flags and the content of every file the library is built from."""
    h = hashlib.sha256()
    h.update(" ".join(HIPCC_FLAGS + LINK_FLAGS).encode())
    for d in DEPS:
        h.update(d.encode())
        with open(os.path.join(CSRC, d), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def built_id():
    try:
        with open(ID_PATH) as f:
            return f.read().strip()
    except OSError:
        return None


def is_stale():
    return not os.path.exists(LIB_PATH) or built_id() != source_id()


def build_extension(force=False, verbose=False):
    """Compile csrc/*.hip into libfractal_hip.so.  Returns the path."""
    if not force and not is_stale():
        return LIB_PATH
    sid = source_id()
    cmd = ([find_hipcc()] + HIPCC_FLAGS + ['-DFR_BUILD_ID="%s"' % sid, "-o", LIB_PATH]
           + [os.path.join(CSRC, s) for s in SOURCES] + LINK_FLAGS)
    if verbose:
        print(" ".join(cmd))
    if os.path.exists(ID_PATH):
        os.remove(ID_PATH)
    subprocess.run(cmd, check=True, cwd=CSRC)
    with open(ID_PATH, "w") as f:
        f.write(sid + "\n")
    return LIB_PATH


if __name__ == "__main__":
    print(build_extension(force=True, verbose=True))
