#!/bin/bash
# rocprofv3 kernel trace of the two-pass render (tile 11) on BASELINE C4, f32 and f64.
# Usage (GPU box, repo root): bash tools/profile_two_pass.sh <tag> [first_cap]
set -u
TAG=${1:-r02}; K1=${2:-128}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof2p_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for P in f32 f64; do
  rocprofv3 --kernel-trace --stats -d "$OUT/$P/stats" -o bench -- python3 "$REPO/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-extras \
      --view julia --iterations 4096 --precision $P --tile 11 --refill $K1,-1 > "$OUT/$P.json" 2> "$OUT/$P.err"
  echo "$P rc=$?"
  python3 "$REPO/tools/rocpd_summary.py" "$OUT/$P" > "$OUT/$P.summary.txt" 2>&1 || true
  cat "$OUT/$P.summary.txt" | head -20
done
