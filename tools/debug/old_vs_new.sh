#!/bin/bash
# The library of the round's start (tools/debug/variants/libfractal_hip_old.so: tools/debug/build_variants.sh builds it from commit b01bede) against the
# current one, bench.py --no-extras per line, alternating processes on ONE box (a box's clock state differs by more than a change).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=$REPO/fractal-renderer_amd/libfractal_hip.so
cp "$LIB" /tmp/lib_new.so
for round in 1 2 3; do
  for v in new old; do
    if [ $v = new ]; then cp /tmp/lib_new.so "$LIB"; else cp "$REPO/tools/debug/variants/libfractal_hip_old.so" "$LIB"; fi
    for args in "--view julia --iterations 4096 --precision f32" "--view julia --iterations 4096" ""; do
      python3 "$REPO/bench.py" --no-extras --no-cpu-baseline --steps 20 --warmup 3 $args > /tmp/b.json 2>/tmp/b.err || { echo "bench failed ($v $args)"; tail -3 /tmp/b.err; cp /tmp/lib_new.so "$LIB"; exit 1; }
      python3 - "$v" "$args" <<'PY'
import json, sys
d = json.loads([l for l in open('/tmp/b.json') if l.startswith('{')][-1])
print("%-4s %-48s ms_per_step %8.3f kernel_ms_avg %8.3f frac %.4f build %s" % (sys.argv[1], sys.argv[2] or "c2 f64", d["ms_per_step"], d["kernel_ms_avg"], d["roofline"]["frac"], d["build_id"]), flush=True)
PY
    done
  done
done
cp /tmp/lib_new.so "$LIB"
