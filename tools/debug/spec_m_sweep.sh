#!/bin/bash
# Block length of the speculative blocks (FR_SPEC_M = 8 / 16 / 32): the library variants under tools/debug/variants
# (tools/debug/build_variants.sh) are copied over the box's scratch copy of the library in turn; bench.py's headline only.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=$REPO/fractal-renderer_amd/libfractal_hip.so
cp "$LIB" /tmp/lib_m16.so
for round in 1; do
  for v in m16 m8 m32 m16; do
    if [ $v = m16 ]; then cp /tmp/lib_m16.so "$LIB"; else cp "$REPO/tools/debug/variants/libfractal_hip_$v.so" "$LIB"; fi
    for args in "" "--precision f32" "--view zoom1e6 --iterations 65536 --steps 2 --warmup 1" "--view julia --iterations 4096 --precision f32" "--view julia --iterations 4096"; do
      python3 "$REPO/bench.py" --no-extras --no-cpu-baseline --steps 20 --warmup 3 $args > /tmp/b.json 2>/tmp/b.err || { echo "bench failed ($v $args)"; tail -3 /tmp/b.err; exit 1; }
      python3 - "$v" "$args" <<'PY'
import json, sys
d = json.loads([l for l in open('/tmp/b.json') if l.startswith('{')][-1])
print("%-4s %-40s ms_per_step %8.3f kernel_ms_avg %8.3f frac %.4f build %s" % (sys.argv[1], sys.argv[2] or "c2 f64", d["ms_per_step"], d["kernel_ms_avg"], d["roofline"]["frac"], d["build_id"]), flush=True)
PY
    done
  done
done
cp /tmp/lib_m16.so "$LIB"
