#!/bin/bash
# the work-queue kernel's episode policy and variants on C4 (GPU box, from the repo root)
cd "${GRAFT_REPO_ROOT:-.}"
python -m pytest tests/test_gpu_kernels_r02.py -x -q -m gpu -k "queue or c4 or kats" 2>&1 | tail -3
run() { python bench.py --view julia --iterations 4096 --no-extras --steps 10 "$@" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['roofline']['frac'],4), d['roofline']['kernel'][:30])"; }
for prec in f32 f64; do
 for pol in "" "8,6" "4,4" "16,8" "8,10" "4,12" "0,16"; do echo "== $prec tile10 refill $pol"; if [ -z "$pol" ]; then run --precision $prec --tile 10; else run --precision $prec --tile 10 --refill $pol; fi; done
 echo "== $prec tile 9"; run --precision $prec --tile 9
done
python tools/queue_trace.py f32 2>&1 | grep -v amdgpu.ids | tail -8
FR_TRACE=1 python tools/e2e_host_time.py 2>&1 | grep -v amdgpu.ids | tail -12
