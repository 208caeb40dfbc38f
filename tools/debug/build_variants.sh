#!/bin/bash
# Builds the library variants the A/B scripts of this directory copy over the box's library (they are git-ignored binaries):
#   libfractal_hip_m8.so / _m32.so   the current sources with -DFR_SPEC_M=8 / 32   (spec_m_sweep.sh)
#   libfractal_hip_old.so            the library of the round's start, commit b01bede, plus a stub for the one entry point
#                                    that did not exist yet (fr_debug_loop_plan; _native.py binds every prototype)  (old_vs_new.sh)
# Run in the build container, from the repo root: bash tools/debug/build_variants.sh
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/tools/debug/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared -pthread"
SRCS="fr_kernels.hip fr_api.hip fr_host.hip fr_multi.hip fr_fern.hip"
mkdir -p "$OUT"
cd "$ROOT/fractal-renderer_amd/csrc"
for M in 8 32; do
  /opt/rocm/bin/hipcc $FLAGS -DFR_SPEC_M=$M -DFR_BUILD_ID="\"specM$M\"" -o "$OUT/libfractal_hip_m$M.so" $SRCS -ldl &
done
OLD=$(mktemp -d)
mkdir -p "$OLD/fractal-renderer_amd/csrc" "$OLD/include"
for f in $SRCS fr_kernels.h fr_ctx.h fr_math.h fr_log2_table.inc; do git -C "$ROOT" show b01bede:fractal-renderer_amd/csrc/$f > "$OLD/fractal-renderer_amd/csrc/$f"; done
git -C "$ROOT" show b01bede:include/fractal_hip.h > "$OLD/include/fractal_hip.h"
python3 - "$OLD/fractal-renderer_amd/csrc/fr_api.hip" <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
s = s.replace("int fr_set_loop_mode(int mode) {", "int fr_debug_loop_plan(const fr_config *, int, uint32_t *, double *, uint32_t *) { return 2; } /* stub: this is the OLD library */\n\nint fr_set_loop_mode(int mode) {", 1)
open(p, "w").write(s)
PY
(cd "$OLD/fractal-renderer_amd/csrc" && /opt/rocm/bin/hipcc $FLAGS -DFR_BUILD_ID="\"old_b01bede\"" -o "$OUT/libfractal_hip_old.so" $SRCS -ldl)
wait
rm -rf "$OLD"
ls -la "$OUT"
