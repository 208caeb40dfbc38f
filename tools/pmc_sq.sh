#!/bin/bash
# Per-kernel SQ counters of one `bench.py --no-extras` invocation, two PMC passes (8 SQ slots each), plus a
# --kernel-trace --stats pass.  Usage (on the GPU box, from the repo root):  bash tools/pmc_sq.sh <tag> [bench args...]
# Writes gpurun_out/pmc_<tag>/summary.txt (copy what is to be judged into profiles/).
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
ARGS="--no-extras --steps 10 --warmup 2 $*"   # the default bench's step counts
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o b -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_stats.json" 2> "$OUT/stats.err"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d "$OUT/pmc1" -o b -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_pmc1.json" 2> "$OUT/pmc1.err"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d "$OUT/pmc2" -o b -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_pmc2.json" 2> "$OUT/pmc2.err"
# the box's own reference (boxes of the pool differ by +-4 % in sustained clock): the same workload without the profiler,
# speculative blocks on (default) and off (--loop-mode 5), back to back
python3 "$REPO/bench.py" $ARGS --no-cpu-baseline > "$OUT/bench_plain.json" 2> "$OUT/plain.err"
python3 "$REPO/bench.py" $ARGS --no-cpu-baseline --loop-mode 5 > "$OUT/bench_nospec.json" 2> "$OUT/nospec.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o b -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_pmcw.json" 2> "$OUT/pmcw.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o b -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_pmcf.json" 2> "$OUT/pmcf.err"
python3 - "$OUT" "$TAG" "$ARGS" > "$OUT/summary.txt" <<'PY'
import glob, json, os, sqlite3, sys
root, tag, args = sys.argv[1], sys.argv[2], sys.argv[3]
print("# rocprofv3 summary '%s': bench.py %s" % (tag, args))
def q(db, sql):
    con = sqlite3.connect(db)
    try:
        return con.execute(sql).fetchall()
    finally:
        con.close()
for sub in ("stats", "pmc1", "pmc2", "pmc_write", "pmc_fetch"):
    dbs = glob.glob(os.path.join(root, sub, "*.db"))
    if not dbs:
        print("== %s: no database" % sub)
        continue
    db = dbs[0]
    if sub == "stats":
        print("== kernel-trace --stats: kernel, calls, avg_us, min_us, max_us")
        for r in q(db, "select name, count(*), avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels "
                       "group by name order by sum(end-start) desc"):
            if "escape" in r[0] or "palette" in r[0] or "fern" in r[0]:
                print("%s, %d, %.1f, %.1f, %.1f" % r)
    else:
        print("== %s: kernel, counter, dispatches, avg_value_per_dispatch" % sub)
        for r in q(db, "select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                       "group by kernel_name, counter_name order by 1, 2"):
            if "escape" in r[0] or "fern" in r[0]:
                print("%s, %s, %d, %.6g" % r)
for name, what in (("bench_stats.json", "bench.py line of the stats pass"),
                   ("bench_plain.json", "same box, no profiler, speculative blocks on (default)"),
                   ("bench_nospec.json", "same box, no profiler, speculative blocks OFF (--loop-mode 5: round 3's loops)")):
    try:
        d = json.loads(open(os.path.join(root, name)).read().strip().splitlines()[-1])
        print("== %s: ms_per_step %.3f kernel_ms_avg %.3f value %.4g frac %.4f kernel %s" % (
            what, d["ms_per_step"], d["kernel_ms_avg"], d["value"], d["roofline"]["frac"], d["roofline"]["kernel"]))
    except Exception as e:  # noqa: BLE001
        print("== %s: no bench line (%r)" % (what, e))
PY
cat "$OUT/summary.txt"
