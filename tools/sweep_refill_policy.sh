#!/bin/bash
# The patch-refill kernel's episode policy on BASELINE C4 (f32 and f64): fr_set_refill_policy(minrun, quit16).
cd ${GRAFT_REPO_ROOT:-.}
run() { python bench.py --view julia --iterations 4096 --no-extras --steps 10 --tile 9 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"; }
for prec in f32 f64; do
 for mr in 8 16 32 48; do for q in 4 6 8 10 12; do echo -n "$prec minrun $mr quit16 $q: "; run --precision $prec --refill $mr,$q; done; done
done
