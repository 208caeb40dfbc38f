#!/usr/bin/env python3
"""Turn the rocprofv3 passes of tools/pmc_sq.sh into the record bench.py attaches to its line:
    python tools/pmc_record.py <key>=<gpurun_out/pmc_TAG dir> ...   ->  profiles/pmc_counters.json
<key> = "<W>x<H>_i<iterations>_<f32|f64>_<view>" (bench.py's name for the configuration).
Per configuration (all render kernels of a launch added up): HBM bytes per launch = WRITE_SIZE + 2 x FETCH_SIZE (KiB ->
bytes; the doubling is MI355X_MICROARCH.md's gfx950 correction), vector / scalar instructions, active cycles
(GRBM_GUI_ACTIVE / 8 XCDs), and
    valu_issue_util = SQ_INSTS_VALU x nominal issue cycles (2 for an f32 render, 4 for f64) / 1024 SIMDs / active cycles
— the share of vector-issue slots the launch used if every vector instruction were of the render's own type (an f64
render issues some f32-rate instructions, so its figure is an upper bound; an f32 render issues some 4-cycle integer,
64-bit and transcendental ones, so its figure is a lower bound).  The build_id of the library that was profiled is
stored: bench.py shows the record only when it matches the library it runs."""
import datetime
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path = os.path.join(ROOT, "profiles", "pmc_counters.json")
try:
    data = json.load(open(out_path))
except (OSError, ValueError):
    data = {}
data["_comment"] = __doc__.split("\n", 1)[1].strip()
for arg in sys.argv[1:]:
    key, d = arg.split("=", 1)
    counters = {}
    kernels = set()
    for line in open(os.path.join(d, "summary.txt")):
        parts = [p.strip() for p in line.rsplit(",", 3)]
        # render kernels only: not bench.py's one COUNT-mode launch (escape_strip_kernel<T, 2, K> / the refill kernel)
        if (len(parts) == 4 and parts[1].isupper() and ("escape_" in parts[0]) and "refill" not in parts[0] and "_v1_" not in parts[0]
                and not re.search(r"escape_strip_kernel<\w+, [12],", parts[0])):
            try:
                counters[parts[1]] = counters.get(parts[1], 0.0) + float(parts[3])
                kernels.add(parts[0].split("(anonymous namespace)::")[-1].split("(")[0])
            except ValueError:
                pass
    bench = json.loads(open(os.path.join(d, "bench_stats.json")).read().strip().splitlines()[-1])
    f32 = "_f32_" in key
    cycles = counters["GRBM_GUI_ACTIVE"] / 8.0
    rec = {
        "build_id": bench["build_id"],
        "date": datetime.date.today().isoformat(),
        "kernels": sorted(kernels),
        "write_size_kib": counters["WRITE_SIZE"], "fetch_size_kib": counters["FETCH_SIZE"],
        "hbm_bytes_per_launch": int(1024 * (counters["WRITE_SIZE"] + 2 * counters["FETCH_SIZE"])),
        "algorithmic_bytes_per_launch": 3 * bench["config"]["per_gpu_pixels"],
        "sq_insts_valu": counters["SQ_INSTS_VALU"], "sq_insts_salu": counters["SQ_INSTS_SALU"],
        "active_cycles": cycles,
        "thread_cycles_valu_per_inst": counters["SQ_THREAD_CYCLES_VALU"] / counters["SQ_INSTS_VALU"],
        "valu_issue_util": counters["SQ_INSTS_VALU"] * (2 if f32 else 4) / 1024.0 / cycles,
        "valu_issue_util_note": "SQ_INSTS_VALU x %d nominal issue cycles / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8); %s" % (
            2 if f32 else 4, "lower bound (some instructions of an f32 render issue at 4-8 cycles)" if f32 else
            "upper bound (some instructions of an f64 render issue at the f32 rate)"),
        "kernel_ms_under_profiler": bench["kernel_ms_avg"],
        "source": os.path.relpath(d, ROOT),
    }
    data[key] = rec
    print(key, json.dumps(rec)[:300])
json.dump(data, open(out_path, "w"), indent=1)
print("wrote", out_path)
