#!/bin/bash
# VERDICT r03 #3's two structural candidates for C4, priced BEFORE building them:
#   (b) "hand-over entries kept in LDS and drained by the same workgroup": what it could save is the hand-over's global
#       stores, the second pass's reads of them, and the second launch.  Measured here: the first pass with its entry stores
#       ablated (FR_DEBUG_ABLATE=1; the image is WRONG, only the time counts), the first pass alone (=3), and the idle gap
#       between the two kernels of a normal render, from the rocprofv3 kernel trace.
#   (a) "two pixels per lane in f32" halves the scalar control per pixel: what it could save is bounded by how much of the
#       first pass's time is NOT vector issue — SQ_ACTIVE_INST_VALU against SQ_BUSY_CYCLES / SQ_WAVE_CYCLES in tools/pmc_sq.sh's
#       passes (profiles/r04_c4_*_rocprofv3.txt).
# Usage (GPU box, repo root):  bash tools/c4_ablation.sh [f32|f64]   ->  gpurun_out/c4_ablation_<prec>.txt
PREC=${1:-f32}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/c4_ablation_$PREC
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
ARGS="--no-extras --steps 10 --warmup 2 --view julia --iterations 4096 --precision $PREC --no-cpu-baseline"
for A in 0 1 3; do
  export FR_DEBUG_ABLATE=$A
  rocprofv3 --kernel-trace --stats -d "$OUT/a$A" -o b -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_a$A.json" 2> "$OUT/a$A.err"
done
unset FR_DEBUG_ABLATE
python3 - "$OUT" "$PREC" > "$REPO/gpurun_out/c4_ablation_$PREC.txt" <<'PY'
import glob, os, sqlite3, sys
root, prec = sys.argv[1], sys.argv[2]
print("# tools/c4_ablation.sh %s: C4 (Julia -0.8+0.156i, 16384^2, 4096 iterations) under rocprofv3 --kernel-trace, 12 launches each" % prec)
print("# FR_DEBUG_ABLATE: 0 = the normal render; 1 = the first pass claims its list slots but does not STORE the entries (wrong image);")
print("#                  3 = that, and the second pass is not launched.  avg / min microseconds per kernel; gap = start of the second")
print("#                  pass minus end of the first pass of the same render")
for a in (0, 1, 3):
    dbs = glob.glob(os.path.join(root, "a%d" % a, "*.db")) + glob.glob(os.path.join(root, "a%d" % a, "*", "*.db"))
    if not dbs:
        print("ablate %d: no database" % a)
        continue
    con = sqlite3.connect(dbs[0])
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    con.close()
    first = [(s, e) for n, s, e in rows if "escape_first_kernel" in n]
    second = [(s, e) for n, s, e in rows if "escape_second_kernel" in n]
    f = lambda v: "avg %.1f min %.1f" % (sum(v) / len(v) / 1e3, min(v) / 1e3) if v else "-"
    line = "ablate %d: first pass x%d %s" % (a, len(first), f([e - s for s, e in first]))
    if second:
        line += " | second pass x%d %s" % (len(second), f([e - s for s, e in second]))
        gaps = []
        for s2, e2 in second:
            prev = [e for s, e in first if e <= s2 + 1000]
            if prev:
                gaps.append(s2 - max(prev))
        line += " | gap %s" % f(gaps)
        both = [e2 - max(s for s, e in first if s <= s2) for s2, e2 in second]
        line += " | first start -> second end %s" % f(both)
    print(line)
PY
cat "$REPO/gpurun_out/c4_ablation_$PREC.txt"
