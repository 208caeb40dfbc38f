#!/bin/bash
# A/B of the colour filter's f32 first stage on one box: 1 = f32 then f64 stage, 2 = f64 stage only
cd ${GRAFT_REPO_ROOT:-.}
run() { python bench.py --no-extras --steps 10 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"; }
for rep in 1 2; do
 for f in 1 2; do
  echo -n "C2 f64 filter $f: "; run --colour-filter $f
  echo -n "C2 f32 filter $f: "; run --colour-filter $f --precision f32
  echo -n "C4 f32 filter $f: "; run --colour-filter $f --view julia --iterations 4096 --precision f32
  echo -n "C4 f64 filter $f: "; run --colour-filter $f --view julia --iterations 4096 --precision f64
 done
done
