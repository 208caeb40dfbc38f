#!/bin/bash
# rocprofv3 kernel trace of BASELINE C4 (large Julia image: the two-pass render by default dispatch), f32 and f64,
# and the per-wave trace of its second pass.
# Usage (GPU box, repo root): bash tools/profile_two_pass.sh <tag>
set -u
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof2p_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for P in f32 f64; do
  rocprofv3 --kernel-trace --stats -d "$OUT/$P/stats" -o bench -- python3 "$REPO/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-extras \
      --view julia --iterations 4096 --precision $P > "$OUT/$P.json" 2> "$OUT/$P.err"
  echo "$P rc=$?"
  python3 "$REPO/tools/rocpd_summary.py" "$OUT/$P" > "$OUT/$P.summary.txt" 2>&1 || true
  head -20 "$OUT/$P.summary.txt"
  python3 "$REPO/tools/queue_trace.py" $P 11 > "$OUT/$P.queue_trace.txt" 2>&1
  echo "trace rc=$?"
done
