#!/bin/bash
# Collect the rocprofv3 evidence for bench.py's default workload on the GPU box:
#   1. --kernel-trace --stats      (per-kernel durations; must agree with bench.py's HIP-event time)
#   2. --pmc WRITE_SIZE            (HBM write bytes)           } separate passes, as
#   3. --pmc FETCH_SIZE            (HBM read bytes, x2 on gfx950) } MI355X_MICROARCH.md prescribes
#   4. --pmc SQ_* VALU counters    (VALU utilisation)
# Usage (from the repo root, on the GPU box):  bash tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 1 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o bench -- python3 "$REPO/bench.py" $ARGS > "$OUT/stats.json" 2> "$OUT/stats.err"
echo "stats rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o bench -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err"
echo "pmc_write rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o bench -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
echo "pmc_fetch rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE -d "$OUT/pmc_sq" -o bench -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_sq.json" 2> "$OUT/pmc_sq.err"
echo "pmc_sq rc=$?"
find "$OUT" -name "*.csv" | head -50
