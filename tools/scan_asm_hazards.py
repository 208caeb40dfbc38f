#!/usr/bin/env python3
"""Static check of the library's gfx950 ISA for the one wait-state rule the compiler cannot apply for us: a memory
instruction written in an asm statement that reads a scalar register which a VECTOR instruction (v_readlane /
v_readfirstlane: a spill restore, a uniform value coming back from the vector unit) wrote fewer than five wait states
earlier (the compiler inserts those wait states in its own code, but does not look inside asm statements).
Usage: python tools/scan_asm_hazards.py [listing.s]    (no argument: compiles fr_kernels.hip with the build's flags)
Exit code 1 if a pattern is found.  Linear look-back (branches ignored): a lint, not a proof."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fractal-renderer_amd", "csrc")


def scan(lines):
    bad = []
    for i, ln in enumerate(lines):
        s = ln.strip()
        if not s.startswith(("global_", "buffer_", "flat_", "scratch_")):
            continue
        regs = set()
        for a, b in re.findall(r"s\[(\d+):(\d+)\]", s):
            regs |= set(range(int(a), int(b) + 1))
        if not regs:
            continue
        waited, j = 0, i - 1
        while j >= 0 and waited < 5:
            t = lines[j].strip()
            j -= 1
            if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
                continue
            if t.startswith("s_nop"):
                waited += int(t.split()[1]) + 1
                continue
            waited += 1
            m = re.match(r"v_(?:readlane|readfirstlane)_b32 s(\d+)", t)
            if m and int(m.group(1)) in regs:
                bad.append((i + 1, s, t))
    return bad


def main(argv=()):
    if len(argv) > 0:
        lines = open(argv[0]).read().split("\n")
    else:
        flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "--cuda-device-only", "-S"]
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "k.s")
            subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-o", out, os.path.join(CSRC, "fr_kernels.hip")], check=True, cwd=CSRC,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            lines = open(out).read().split("\n")
    bad = scan(lines)
    for ln, use, write in bad:
        print("line %d: %s   <-   %s" % (ln, use, write))
    print("%d memory instructions read a scalar register a vector instruction wrote < 5 wait states before" % len(bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
