#!/usr/bin/env python3
"""the CPU path (oracle's row-parallel driver, the stand-in for the reference's rayon
loop, which cannot be built here) timed IN FULL on C1 and C2 next to the GPU, with a byte comparison
of the complete images.  Prints one line per config.  Runs on the GPU box."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fractal_renderer_amd as fr  # noqa: E402
import oracle_lib as O  # noqa: E402

fr.init(0)
threads = min(16, len(os.sched_getaffinity(0)))
model = next((ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")), "?")
print("host: %s, %d threads used of %d online" % (model, threads, os.cpu_count()))
CASES = {
    "C1 3000x3000 zoom 1e6 i=1024": O.cli_config(3000, 3000, iterations=1024, scale=(1e6, 1e6),
                                                 pos=(-0.7436447860, 0.1318252536)),
    "C2 16384x16384 default i=1024": O.cli_config(16384, 16384, iterations=1024),
}
for name, ocfg in CASES.items():
    cfg = fr.Config.from_buffer_copy(bytes(ocfg))
    fr.get_image(cfg)
    t0 = time.perf_counter()
    gpu = fr.get_image(cfg)
    t_gpu = time.perf_counter() - t0
    total, _ = fr.count_iterations(cfg)
    for mode, tag in ((O.LOG2_LIBM, "libm log2 (the reference's)"), (O.LOG2_SOFT, "soft log2")):
        O.set_log2_mode(mode)
        t0 = time.perf_counter()
        cpu = O.get_image(ocfg, threads=threads)
        t_cpu = time.perf_counter() - t0
        O.set_log2_mode(O.LOG2_LIBM)
        diff = int((cpu != gpu).sum())
        print("%s | %s: CPU %.2f s = %.3e px-it/s | GPU end-to-end (host buffer) %.1f ms = %.3e px-it/s | "
              "differing bytes over the whole image: %d of %d" % (name, tag, t_cpu, total / t_cpu, t_gpu * 1e3,
                                                                    total / t_gpu, diff, cpu.size), flush=True)
        del cpu
