#!/bin/bash
# SQ_INSTS_VALU / SQ_WAVES of one bench.py invocation (no --kernel-trace domains besides kernel): per-kernel VALU counts.
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d "$OUT" -o b -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/err.log"
python3 - "$OUT" <<'PY'
import sqlite3,glob,sys
db=glob.glob(sys.argv[1]+"/*.db")[0]
con=sqlite3.connect(db)
rows=con.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
d={}
for k,c,v in rows: d.setdefault(k,{})[c]=v
for k,v in d.items():
    if 'escape' in k and v.get('SQ_WAVES'):
        w=v['SQ_WAVES']; print(k[:70], "waves %.0f" % w, " ".join("%s/wave %.1f" % (c[8:], v[c]/w) for c in sorted(v) if c!='SQ_WAVES'))
PY
