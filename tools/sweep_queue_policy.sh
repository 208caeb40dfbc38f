cd $GRAFT_REPO_ROOT
run() { python bench.py --view julia --iterations 4096 --no-extras --steps 10 "$@" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['roofline']['frac'],4), d['roofline']['kernel'][:30])"; }
for prec in f32 f64; do
 for b in 1 2 4; do echo "== $prec batch $b"; FR_QUEUE_BATCH=$b run --precision $prec; done
 for pol in "8,6" "4,4" "16,8" "0,3" "8,3" "24,8"; do echo "== $prec batch 1 refill $pol"; FR_QUEUE_BATCH=1 run --precision $prec --refill $pol; done
 echo "== $prec tile 9"; run --precision $prec --tile 9
done
