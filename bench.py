#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on BASELINE.json's configs, on N MI355X of one node.

Metric: pixel-iterations/s: Σ over pixels of EXECUTED loop iterations ÷ time, the
Σ being an exact integer counted on the device outside the timed region (and cross-checked against
the CPU oracle by the cpu_baseline leg).

Workloads (--workload):
    c2    (default) Mandelbrot 16384², default view (-x -0.6 -y 0 -s 0.4), max_iter 1024, fp64 — the
          config BASELINE.json's metric is quoted on.  At N GPUs the SAME image is split
          row-block-cyclically over the ranks ("scaling": "strong": the metric reads "at 16384², 1/2/4/8 GPU").
    c5    Mandelbrot 65536², same view and cap (BASELINE C5: "65536² tiled across 8 GPUs"), any N.
    weak  per-GPU pixel count fixed at 16384²: W = H = round(16384·√N).
A step = one pass of the hot path over the whole image: every rank renders its row blocks into HBM
(inputs are just the Config; outputs stay resident in HBM) and, for N > 1, every finished block travels
to the first GPU over xGMI — straight into its place in the full image — while the next ones render
(the path's one real exchange step, north_star: "final RCCL gather over xGMI").

How N > 1 runs:
  * under a launcher (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`, what the
    driver does): one process per GPU, the gather is torch.distributed point-to-point = RCCL;
  * plain `python3 bench.py --gpus N`: ONE process drives all N GPUs through the library's own
    multi-device entry points (fr_init_devices + fr_render_rgb8_multi_device: one host thread and one
    set of streams per GPU, gather by grouped ncclSend/ncclRecv; --gather peer for xGMI peer DMA), and
    also times the host-buffer variant (every GPU DMAs its blocks to the caller's buffer over its own
    PCIe link).  On a box with fewer than N GPUs it prints a JSON line with an "error" field and exits 0.

One JSON line on stdout (rank 0).  `roofline` prices the escape+colour kernel against the gfx950
VECTOR peak of the arithmetic type (this path is neither HBM- nor MFMA-bound);
`cpu_baseline` is the oracle's row-parallel driver timed on this box's host cores.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# peaks from /opt/skills/guides/MI355X_MICROARCH.md (chip-level parameters) and SURVEY.md §8d
FP64_VECTOR_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 16 lanes x 2 flop(FMA) x 2.4 GHz
FP32_VECTOR_PEAK_TFLOPS = 157.3
FLOPS_PER_ITERATION = 10  # as written in the reference: calc/src/lib.rs:88-89,95,103-104

VIEWS = {
    # name: (algo, pos, scale, julia_set)
    "default": ("mandelbrot", (-0.6, 0.0), 0.4, None),
    "zoom1e6": ("mandelbrot", (-0.7436447860, 0.1318252536), 1e6, None),
    "julia": ("julia", (0.0, 0.0), 0.4, (-0.8, 0.156)),
    # gui_latency only: a filled Julia set (the basilica) and a thin dust of a dozen iterations per pixel
    "filled_julia": ("julia", (0.0, 0.0), 0.4, (-1.0, 1e-9)),
    "thin_dust": ("julia", (0.0, 0.0), 0.4, (0.4, 0.4)),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["c2", "c5", "weak"], default="c2")
    ap.add_argument("--size", type=int, default=0, help="image edge (overrides the workload's)")
    ap.add_argument("--iterations", type=int, default=1024)
    ap.add_argument("--precision", choices=["f64", "f32"], default="f64")
    ap.add_argument("--view", choices=sorted(VIEWS), default="default")
    ap.add_argument("--block-rows", type=int, default=256)
    ap.add_argument("--gather", choices=["rccl", "peer"], default="rccl",
                    help="single-process N > 1: how finished blocks reach the first GPU")
    ap.add_argument("--root-share", type=int, default=1, choices=[0, 1, 2, 4],
                    help="N > 1: the share of ITS row blocks the sink (rank 0 / the first device) renders: 1 = all (plain cyclic "
                         "dealing, default), 2 / 4 = a half / a quarter (the rest is dealt to the other ranks), 0 = none: it only "
                         "receives.  Same bytes; what the first real multi-GPU run should try when rank 0 shows the longest step")
    ap.add_argument("--logical", action="store_true",
                    help="single-process N > 1 on fewer GPUs: N logical devices on GPU 0 (a plumbing check, "
                         "never a scaling number)")
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--refill", default="", help="tuning: minrun,quit16 of the refilling kernel")
    ap.add_argument("--loop-mode", type=int, default=-1, help="tuning: force the orbit loop form (0, 2, 4)")
    ap.add_argument("--no-colour-filter", action="store_true", help="tuning: always the f64 software log2")
    ap.add_argument("--colour-filter", type=int, default=1, help="tuning: 1 = f32 then f64 stage (default), 2 = f64 stage only")
    ap.add_argument("--cycle-shortcut", action="store_true",
                    help="measure with the exact periodicity shortcut on (never the headline: it skips iterations)")
    ap.add_argument("--force-blocks", action="store_true",
                    help="N=1 only: render block by block like a rank of an N>1 run does (tuning of --block-rows)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only (profiling runs)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = every core this process may use")
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)  # the measured run of a plain --gpus N (see supervise)
    ap.add_argument("--child-timeout", type=float, default=float(os.environ.get("FR_BENCH_CHILD_TIMEOUT", "900")),
                    help="plain --gpus N: seconds a measured child process may take before it is killed and the peer-DMA "
                         "gather is tried in a fresh one")
    return ap.parse_args()


def make_config(fr, view, edge, iterations):
    """CLI-default Config (src/lib.rs:34-226: limit 65536, stable_limit 2, exposure 5, inside,
    smooth, default colours) for the chosen view."""
    algo, pos, scale, julia = VIEWS[view]
    cfg = fr.Config.new(fr.Algo.Julia if algo == "julia" else fr.Algo.Mandelbrot)
    cfg.width = cfg.height = edge
    cfg.iterations = iterations
    cfg.exposure = 5.0
    cfg.pos.re, cfg.pos.im = pos
    cfg.scale.re = cfg.scale.im = scale
    if julia:
        cfg.julia_set.re, cfg.julia_set.im = julia
    return cfg


def usable_cores():
    """Cores this process can really use: the affinity mask, capped by the cgroup's CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(math.ceil(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return ""


def cpu_baseline(cfg_bytes, precision, threads, sample):
    """Time the CPU oracle (test infrastructure, used here ONLY as the reported baseline) on every
    `sample`-th pixel in x and y of the same workload (1 = the whole image).  libm log2: what the
    reference's f64::log2 calls.  Returns (rate dict, sampled colours, sampled Σ)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    ocfg = O.Config.from_buffer_copy(cfg_bytes)
    O.set_log2_mode(O.LOG2_LIBM)
    t0 = time.perf_counter()
    total, npx, colours = O.sample_image(ocfg, sample, sample, precision, threads)
    dt = time.perf_counter() - t0
    return {"value": total / dt, "seconds": dt, "pixels": npx, "pixel_iterations": total, "threads": threads}, colours, total


def loop_mix(kernel_name):
    """Vector instructions the kernel's orbit loop issues per iteration AT BEST (counted in the
    ISA: profiles/r01_inner_loop_isa.txt, profiles/r03_c4_first_pass_classes.txt) and what that mix can reach of the nominal
    vector peak if every one of them issued at the arithmetic type's rate and every lane was busy:
    10 flops (the reference's count) / (instructions x 1 lane-slot each) / 2 flops per FMA slot."""
    # round 4: quiet waves (no lane near the limit for 16 iterations: interiors, where C2 / C3 / C5 do 97 % of their work)
    # run speculative blocks of 16 unchecked iterations with ONE distance add + compare at the end and the block's start
    # state kept in a second register set (rolled back and re-run with checks if the test fails): (16 x 6 + 2) / 16
    spec = ("; quiet waves: speculative blocks of 16 unchecked iterations, one test at the end, start state kept for "
            "rollback = (96 + 2) / 16 = 6.125 (the figure used here)")
    if "first_kernel" in kernel_name or "second_kernel" in kernel_name or "queue_kernel" in kernel_name:
        f32 = "<float" in kernel_name
        if "speculative" in kernel_name:  # the first pass's form whose later episodes speculate (launched unless the view's statistics say nothing stays)
            n, what = 6.125, ("scaled form; while lanes leave: unchecked blocks of 4 iterations with a |z|^2 <= T test and freeze "
                              "per block = 26 / 4 (f32, counting per lane: 27 / 4)" + spec)
        else:  # the plain form (C4: no sampled pixel at the cap, mean under 96 iterations: nothing would ever speculate)
            n, what = (6.75, "scaled form in unchecked blocks of 4 iterations with a per-lane count: 24 arithmetic + |z|^2 <= T test + "
                             "count + freeze per block = 27 / 4") if f32 else (
                       6.5, "scaled form in unchecked blocks of 4 iterations: 24 arithmetic + |z|^2 <= T test + freeze per block = 26 / 4 "
                            "(first episode, counting per lane: 27 / 4)")
    elif "strip_kernel" in kernel_name or "refill_kernel" in kernel_name or "escape kernels" in kernel_name:
        n, what = 6.125, ("scaled form X = 2re, Y = 2im, A = X^2, B = Y^2: 6 per iteration; while lanes leave: one distance add "
                          "and one compare per block of 4 = 26 / 4 (the reference as written: 8 arithmetic + 1 compare = 9)" + spec)
    else:
        n, what = 9.0, "the reference's iteration as written: 8 arithmetic + 1 compare"
    return n, what


def roofline_block(prec_name, launch_px_it, kernel_ms, pixels, kernel_name, traffic):
    peak = FP32_VECTOR_PEAK_TFLOPS if prec_name == "f32" else FP64_VECTOR_PEAK_TFLOPS
    achieved = FLOPS_PER_ITERATION * launch_px_it / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
    n_instr, mix_what = loop_mix(kernel_name)
    mix_ceiling = FLOPS_PER_ITERATION / (2.0 * n_instr)
    rate = launch_px_it / (kernel_ms * 1e-3) if kernel_ms > 0 else 0.0
    return {
        "loop_vector_instructions_per_iteration": n_instr,
        "loop_mix": mix_what,
        "mix_ceiling_frac": mix_ceiling,
        "frac_of_mix_ceiling": achieved / peak / mix_ceiling,
        "fraction_of_attainable": 8.0 * rate / (peak / 2.0 * 1e12),
        "fraction_of_attainable_note": "BASELINE.md §2's second figure, 8 x rate / (peak / 2): it prices the loop at 8 non-fused "
                                       "vector operations per iteration; this kernel issues %.2f (%s), which is why the figure can "
                                       "exceed 1 — no iteration is skipped: the executed-iteration sum and every output byte equal the "
                                       "CPU oracle's (cpu_baseline.gpu_*_identical_on_sample)" % (n_instr, mix_what.split(":")[0]),
        "bound": "valu_f64" if prec_name == "f64" else "valu_f32",
        "achieved": achieved,
        "peak": peak,
        "unit": "TFLOP/s",
        "frac": achieved / peak,
        "traffic": traffic,
        "kernel": kernel_name,
        "kernel_ms_avg": kernel_ms,
        "algorithmic_flops_per_launch": FLOPS_PER_ITERATION * launch_px_it,
        "hbm_check": {"algorithmic_bytes_per_launch": 3 * pixels,
                      "achieved_GBps": 3 * pixels / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0,
                      "peak_GBps": 8000.0},
    }


def pmc_record(cfg, prec_name, view, build_id):
    """The committed rocprofv3 PMC record of this exact configuration (profiles/pmc_counters.json, written by
    tools/pmc_record.py from the separate --pmc passes of tools/pmc_sq.sh): HBM bytes per launch and vector-issue
    utilisation.  It is a MEASUREMENT OF ANOTHER RUN, so it is attached only when it was taken with the library
    that is running now (same build_id) and the line says where it comes from."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_counters.json")) as f:
            rec = json.load(f).get("%dx%d_i%d_%s_%s" % (cfg.width, cfg.height, cfg.iterations, prec_name, view))
    except (OSError, ValueError):
        return None
    if not rec or rec.get("build_id") != build_id:
        return None
    return rec


def roofline_with_pmc(block, rec):
    """roofline block + what the committed counters of the same build say; traffic stays null without them"""
    if rec is None:
        block["traffic"] = None
        block["traffic_source"] = "none for this build (profiles/pmc_counters.json holds no record with this build_id)"
        return block
    block["traffic"] = rec["hbm_bytes_per_launch"]
    block["traffic_source"] = "profiles/pmc_counters.json (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes, %s, build_id %s; not measured in this run)" % (
        rec.get("date", "?"), rec["build_id"])
    block["valu_issue_util"] = rec.get("valu_issue_util")
    block["valu_issue_util_note"] = rec.get("valu_issue_util_note")
    if rec.get("active_cycles") and rec.get("kernel_ms_under_profiler"):
        # the clock the render kernels actually ran at in the profiled run: active cycles / their duration (nominal 2.4 GHz)
        block["measured_clock_ghz"] = rec["active_cycles"] / (rec["kernel_ms_under_profiler"] * 1e-3) / 1e9
        block["measured_clock_note"] = "GRBM_GUI_ACTIVE / 8 XCDs / the kernels' duration in the same profiled run (%s); peak is quoted at 2.4 GHz" % rec.get("source", "?")
    if rec.get("sq_insts_valu"):
        block["vector_instructions_per_launch"] = rec["sq_insts_valu"]
        block["loop_floor_vector_instructions_per_launch"] = block["algorithmic_flops_per_launch"] / FLOPS_PER_ITERATION / 64.0 * block[
            "loop_vector_instructions_per_iteration"]
    return block


ROOFLINE_NOTE = ("bound = VECTOR issue rate of the arithmetic type (no MFMA; not HBM: 3 B/pixel written once, see "
                 "hbm_check); achieved = 10 flops (the reference's count per iteration) x executed pixel-iterations "
                 "per launch / the kernel's HIP-event duration")


class SingleGpu:
    """One GPU, one process: the device-pointer API on torch's current stream."""

    def __init__(self, torch, fr, lib, native, device):
        self.torch, self.fr, self.lib, self.native, self.device = torch, fr, lib, native, device
        self.stream = torch.cuda.current_stream(device)
        self.image = None

    def buffer(self, nbytes):
        if self.image is None or self.image.numel() < nbytes:
            self.image = None
            self.image = self.torch.empty(nbytes, dtype=self.torch.uint8, device=self.device)
        return self.image

    def render(self, cfg, prec):
        need = 3 * cfg.width * cfg.height
        buf = self.buffer(need)
        self.native.check(self.lib.fr_render_rows_rgb8_device(C.byref(cfg), int(prec), 0, cfg.height, buf.data_ptr(),
                                                              need, self.stream.cuda_stream))
        return buf[:need].view(cfg.height, cfg.width, 3)

    def measure(self, cfg, prec, steps, warmup):
        """W untimed + K timed renders of the whole image in one launch each.  Returns timings, the exact
        executed-iteration sum and the name of the kernel that ran."""
        torch, lib, native = self.torch, self.lib, self.native
        native.check(lib.fr_set_profiling(1))
        for _ in range(warmup):
            self.render(cfg, prec)
        torch.cuda.synchronize(self.device)
        kernel_ms = []
        t0 = time.perf_counter()
        for _ in range(steps):
            img = self.render(cfg, prec)
            ms = C.c_float(0)
            native.check(lib.fr_last_kernel_ms(C.byref(ms)))  # HIP events on the launch stream
            kernel_ms.append(ms.value)
        torch.cuda.synchronize(self.device)
        dt = time.perf_counter() - t0
        name = C.create_string_buffer(160)
        native.check(lib.fr_last_kernel_name(name, len(name)))
        total, _ = self.fr.count_iterations(cfg, 0, cfg.height, 1, 1, prec)
        return {"dt": dt, "ms_per_step": dt / steps * 1e3, "kernel_ms": sum(kernel_ms) / len(kernel_ms),
                "total": total, "kernel": name.value.decode(), "image": img}


def other_config_line(sg, fr, name, view, iterations, prec_name, steps, warmup, edge=16384, cpu_compare=False):
    prec = fr.Precision.F32 if prec_name == "f32" else fr.Precision.F64
    cfg = make_config(fr, view, edge, iterations)
    m = sg.measure(cfg, prec, steps, warmup)
    pixels = cfg.width * cfg.height
    return {
        "_compare": (cfg, prec, m["total"]) if cpu_compare else None,  # resolved by whole_image_cpu_compare, after the timed legs
        "workload": "%s %dx%d max_iter=%d %s view=%s (BASELINE %s)" % (VIEWS[view][0], edge, edge, iterations, prec_name, view, name),
        "value": m["total"] * steps / m["dt"], "unit": "pixel-iterations/s", "steps": steps, "warmup": warmup,
        "ms_per_step": m["ms_per_step"], "dtype": prec_name, "pixel_iterations_per_image": m["total"],
        "mean_iterations_per_pixel": m["total"] / pixels,
        "roofline": roofline_with_pmc(roofline_block(prec_name, m["total"], m["kernel_ms"], pixels, m["kernel"] + " (fused coordinate map + "
                                                     "orbit loop + colour map)", None), pmc_record(cfg, prec_name, view, fr.build_id())),
    }


def whole_image_cpu_compare(sg, line):
    """The WHOLE image of a measured config on the host (C4: 1.2e10 pixel-iterations, a second or two of CPU), byte for
    byte against the GPU's — after the timed legs, so that the device does not idle (and drop its clock) between them."""
    job = line.pop("_compare", None)
    if not job:
        return
    cfg, prec, total = job
    img = sg.render(cfg, prec)
    sg.torch.cuda.synchronize(sg.device)
    info, colours, cpu_total = cpu_baseline(bytes(cfg), int(prec), usable_cores(), 1)
    line.update({"cpu_bytes_identical": bool((img.cpu().numpy() == colours).all()),
                 "cpu_iteration_sum_identical": bool(cpu_total == total),
                 "cpu_seconds": info["seconds"], "cpu_threads": info["threads"]})


def gui_latency(fr, lib, native):
    """SURVEY.md §8 f2, driver-timed: what a GUI-shaped caller sees — the reference's render thread
    calls get_image for every redraw (src/gui.rs:56-82) and hands RGBA to egui (:71-72); `S` starts a 2x screenshot
    on another thread while redraws go on (:322-326).  Per frame shape: the FIRST call of the process for that shape,
    then median and 95th percentile of 50 calls of the host-buffer entry point into a buffer that exists (kernel +
    D2H + the call's own overhead, milliseconds), RGB and RGBA; then render + screenshot on two threads at once."""
    import threading

    import numpy as np

    frames = [("750x500 i=50 (CLI defaults, src/lib.rs:34-50)", 750, 500, 50), ("1500x1000 i=50 (the 2x screenshot of it)", 1500, 1000, 50),
              ("1920x1080 i=1024", 1920, 1080, 1024), ("3840x2160 i=1024", 3840, 2160, 1024)]
    out = {}
    native.check(lib.fr_set_profiling(1))  # fr_last_kernel_name: which kernel a frame's render launched
    kname = C.create_string_buffer(256)

    def last_kernel():
        native.check(lib.fr_last_kernel_name(kname, len(kname)))
        return kname.value.decode()

    # the Python side of the two entry points once, on a 16 x 16 frame of its own (another shape, another view): what the
    # first_call_ms of the small frames measures is the LIBRARY's first call for a shape — ctypes' own first-call work
    # (argument conversion set-up, ~50 us) is a visible share of a 0.09 ms frame
    tiny = make_config(fr, "default", 16, 50)
    fr.get_image_rows(tiny, 0, 16, fr.Precision.F64, out=np.ones((16, 16, 3), dtype=np.uint8))
    fr.get_image_rgba(tiny, fr.Precision.F64, out=np.ones((16, 16, 4), dtype=np.uint8))
    for view in ("default", "julia", "filled_julia", "thin_dust"):
        rows = {}
        for label, w, h, it in (frames if view in ("default", "julia") else frames[2:]):
            if view == "thin_dust":
                it, label = 256, label.replace("i=1024", "i=256")
            cfg = make_config(fr, view, 16, it)
            cfg.width, cfg.height = w, h
            rec = {}
            for fmt, call, ch in (("rgb", lambda b: fr.get_image_rows(cfg, 0, h, fr.Precision.F64, out=b), 3),
                                  ("rgba", lambda b: fr.get_image_rgba(cfg, fr.Precision.F64, out=b), 4)):
                buf = np.zeros((h, w, ch), dtype=np.uint8)
                buf.fill(1)  # a buffer that EXISTS (np.zeros maps no pages: its faults were most of a 4K first call)
                t0 = time.perf_counter()
                call(buf)
                first = (time.perf_counter() - t0) * 1e3
                k_first = last_kernel()
                ts = []
                for _ in range(50):
                    t0 = time.perf_counter()
                    call(buf)
                    ts.append((time.perf_counter() - t0) * 1e3)
                second = ts[0]
                ts.sort()
                # kernel_first_frame: dispatched by size (nothing is known about the view yet; its sample runs behind the render);
                # kernel_steady: dispatched from the view's own statistics, from the second frame on
                rec[fmt] = {"first_call_ms": first, "second_call_ms": second, "median_ms": ts[25], "p95_ms": ts[47],
                            "kernel_first_frame": k_first, "kernel_steady": last_kernel()}
            rows[label] = rec
        if view not in ("default", "julia"):
            out["julia -1 (filled: the basilica)" if view == "filled_julia" else "julia 0.4+0.4i (a thin dust, ~10 iterations a pixel)"] = rows
            continue
        # render thread + screenshot thread (2x) at once: 30 redraws of 750x500 while 3 screenshots of 1500x1000 render
        small = make_config(fr, view, 16, 50)
        small.width, small.height = 750, 500
        big = make_config(fr, view, 16, 50)
        big.width, big.height = 1500, 1000
        sbuf = np.ones((500, 750, 3), dtype=np.uint8)
        bbuf = np.ones((1000, 1500, 3), dtype=np.uint8)
        redraw, shots = [], []

        def render_thread():
            for _ in range(30):
                t0 = time.perf_counter()
                fr.get_image_rows(small, 0, 500, fr.Precision.F64, out=sbuf)
                redraw.append((time.perf_counter() - t0) * 1e3)

        def screenshot_thread():
            for _ in range(3):
                t0 = time.perf_counter()
                fr.get_image_rows(big, 0, 1000, fr.Precision.F64, out=bbuf)
                shots.append((time.perf_counter() - t0) * 1e3)

        th = [threading.Thread(target=render_thread), threading.Thread(target=screenshot_thread)]
        [t.start() for t in th]
        [t.join() for t in th]
        redraw.sort()
        rows["render thread (750x500) beside the screenshot thread (1500x1000), two host threads"] = {
            "redraw_median_ms": redraw[len(redraw) // 2], "redraw_max_ms": redraw[-1], "screenshot_median_ms": sorted(shots)[1]}
        out["mandelbrot default view" if view == "default" else "julia -0.8+0.156i (C4's view)"] = rows
    native.check(lib.fr_set_profiling(0))
    out["note"] = ("fr_render_rows_rgb8 / fr_render_rows_rgba8 into a resident host buffer, f64, wall time of the call (kernel + D2H "
                   "+ call overhead); first_call_ms = the first call of this process for that frame shape (the Python wrappers "
                   "have been called once before, on a 16 x 16 frame)")
    return out


def run_single(args, torch, fr, lib, native):
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    fr.init(0)
    prec = fr.Precision.F32 if args.precision == "f32" else fr.Precision.F64
    edge = args.size or (65536 if args.workload == "c5" else 16384)
    cfg = make_config(fr, args.view, edge, args.iterations)
    pixels = cfg.width * cfg.height
    sg = SingleGpu(torch, fr, lib, native, device)
    is_c2 = (edge == 16384 and args.iterations == 1024 and args.view == "default" and args.precision == "f64")
    label = "C2" if is_c2 else "C5's image on one GPU" if (edge == 65536 and args.view == "default") else "variant"

    if args.force_blocks:
        from fractal_renderer_amd import partition as P

        renderer = P.DistributedRenderer(cfg, prec, args.block_rows, device=device, force_blocks=True)
        for _ in range(args.warmup):
            renderer.render()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            img = renderer.render()
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        total, _ = fr.count_iterations(cfg, 0, cfg.height, 1, 1, prec)
        m = {"dt": dt, "ms_per_step": dt / args.steps * 1e3, "kernel_ms": dt / args.steps * 1e3, "total": total,
             "kernel": "block-by-block launches (tuning run)", "image": img}
    else:
        m = sg.measure(cfg, prec, args.steps, args.warmup)
    total, img = m["total"], m["image"]
    rate = total * args.steps / m["dt"]
    out = {
        "metric": "pixel_iterations_per_sec",
        "value": rate,
        "unit": "pixel-iterations/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": m["ms_per_step"],
        "higher_is_better": True,
        "scaling": "strong" if args.workload != "weak" else "weak",
        "vs_baseline": None,
        "dtype": args.precision,
        "data": "synthetic (deterministic: the image is a pure function of the Config)",
        "config": {
            "workload": "%s %dx%d max_iter=%d %s view=%s (BASELINE %s)" % (
                VIEWS[args.view][0], cfg.width, cfg.height, cfg.iterations, args.precision, args.view, label),
            "per_gpu_pixels": pixels,
            "partition": "one launch, whole image",
            "exchange": "none",
        },
        "mpixels_per_sec": pixels * args.steps / m["dt"] / 1e6,
        "pixel_iterations_per_image": total,
        "kernel_ms_avg": m["kernel_ms"],
        "build_id": fr.build_id(),
        "roofline": dict(roofline_with_pmc(roofline_block(args.precision, total, m["kernel_ms"], pixels,
                                                          m["kernel"] + " (fused coordinate map + orbit loop + colour map)", None),
                                           pmc_record(cfg, args.precision, args.view, fr.build_id())), note=ROOFLINE_NOTE),
    }
    if args.no_extras:
        print(json.dumps(out), flush=True)
        return

    # The box's own reference, right behind the timed steps (boxes of the pool differ by +-4 % in sustained clock, so a line
    # read on its own cannot say what a change bought): the same workload with the speculative long blocks switched off
    # (fr_set_loop_mode(5): round 3's loops).  Not part of `value`.
    if args.loop_mode == -1 and args.tile == 0 and not args.force_blocks:
        # (measure() renders into the same buffer; no second 805 MB allocation for the comparison: two sums of the bytes)
        flat = img.reshape(-1)
        on_sums = (int(flat.sum(dtype=torch.int64)), int(flat[1::2].sum(dtype=torch.int64)))
        # reproduction aid: the second 805 MB device allocation, alive across the measurement below, behind which the first
        # two-band host frame of the process stalled 25-30 ms
        clone_probe = img.clone() if os.environ.get("FR_BENCH_DEBUG_CLONE") == "1" else None
        native.check(lib.fr_set_loop_mode(5))
        try:
            off = sg.measure(cfg, prec, max(3, args.steps // 2), 1)
        finally:
            native.check(lib.fr_set_loop_mode(args.loop_mode))
        del clone_probe
        out["roofline"]["same_box_without_speculative_blocks"] = {
            "kernel_ms_avg": off["kernel_ms"], "frac": FLOPS_PER_ITERATION * total / (off["kernel_ms"] * 1e-3) / 1e12 / (
                FP32_VECTOR_PEAK_TFLOPS if args.precision == "f32" else FP64_VECTOR_PEAK_TFLOPS),
            "speculative_blocks_gain": 1.0 - m["kernel_ms"] / off["kernel_ms"],
            "byte_sums_identical": (int(off["image"].reshape(-1).sum(dtype=torch.int64)),
                                    int(off["image"].reshape(-1)[1::2].sum(dtype=torch.int64))) == on_sums,
            "note": "fr_set_loop_mode(5) — the loops of round 3, same process, same box, measured right behind the timed steps; not "
                    "part of `value`",
        }
        img = sg.render(cfg, prec)  # (the reference image of the checks below: rendered with the default loops again)

    ref = img.clone()
    # The other millisecond-scale BASELINE configs right behind the headline, while the device is in the state the
    # headline was measured in: seconds of sustained load (the host-path legs, C3, C5 below) leave it at a lower clock for
    # a while, and C4 read 8 % slower behind them than behind an idle period (DVFS; MI355X_MICROARCH.md).
    if is_c2:
        out["other_configs"] = {
            "C4": other_config_line(sg, fr, "C4", "julia", 4096, "f32", 10, 2, cpu_compare=not args.no_cpu_baseline),
            "C4_f64": other_config_line(sg, fr, "C4 in f64", "julia", 4096, "f64", 10, 2, cpu_compare=not args.no_cpu_baseline),
            "C2_f32": other_config_line(sg, fr, "C2 in f32", "default", 1024, "f32", 10, 2),
        }
        for line in out["other_configs"].values():
            whole_image_cpu_compare(sg, line)
        img = sg.render(cfg, prec)  # the shared device buffer held the other configs meanwhile
        torch.cuda.synchronize(device)
    # t three ways: kernel only = kernel_ms_avg above; the drop-in call as the reference's
    # caller sees it, into a host buffer that already exists (a GUI re-rendering), and into a FRESH one
    # (get_image returns a new Vec every call, src/lib.rs:266-267): kernel + PCIe D2H, never `value`
    import numpy as np

    def host_call(buf):
        th = time.perf_counter()
        fr.get_image_rows(cfg, 0, cfg.height, prec, out=buf)
        return (time.perf_counter() - th) * 1e3

    hbuf = np.empty((cfg.height, cfg.width, 3), dtype=np.uint8)
    host_call(hbuf)
    resident = min(host_call(hbuf) for _ in range(3))
    host_ok = bool((hbuf[::97] == ref[::97].cpu().numpy()).all())
    del hbuf
    fresh = []
    for _ in range(3):
        fbuf = np.empty((cfg.height, cfg.width, 3), dtype=np.uint8)  # never touched: no pages yet
        fresh.append(host_call(fbuf))
        del fbuf
    out["end_to_end"] = {
        "kernel_only_ms": m["kernel_ms"],
        "resident_host_buffer_ms": resident,
        "fresh_host_buffer_ms": min(fresh),
        "fresh_host_buffer_ms_all": fresh,
        "resident_value": total / (resident * 1e-3),
        "fresh_value": total / (min(fresh) * 1e-3),
        "bytes_identical_to_device_image_on_sampled_rows": host_ok,
        "note": "fr_render_rows_rgb8 into a host buffer (kernel + D2H over PCIe, chunk-wise first touch + pin + DMA "
                "overlapped with the rendering); the fresh-buffer time is what a drop-in get_image costs; not `value`",
    }

    # Extra, never the headline: the same steps with the exact periodicity shortcut on (bit-identical
    # output, but periodic orbits are fast-forwarded instead of iterated, so it is not a roofline number)
    if not args.cycle_shortcut:
        native.check(lib.fr_set_cycle_shortcut(1))
        sc = sg.measure(cfg, prec, max(2, args.steps // 2), 1)
        out["exact_cycle_shortcut"] = {
            "ms_per_step": sc["ms_per_step"], "value": total / (sc["ms_per_step"] * 1e-3),
            "bytes_identical_to_plain_loop": bool(torch.equal(ref, sc["image"])), "kernel": sc["kernel"],
            "note": "fr_set_cycle_shortcut(1): orbits that return bitwise to an earlier state are fast-forwarded to "
                    "the cap; same bytes, fewer iterations executed; off by default, excluded from `value` and `roofline`"}
        native.check(lib.fr_set_cycle_shortcut(0))

    # the other single-GPU BASELINE configs, driver-timed in the same run (C3 is ~1.2 s a step: 2 steps)
    if is_c2:
        out["gui_latency"] = gui_latency(fr, lib, native)
        out["other_configs"]["C3"] = other_config_line(sg, fr, "C3", "zoom1e6", 65536, "f64", 2, 1)
        out["other_configs"]["C3"].pop("_compare", None)
        # C5's image (65536^2, 12.9 GB) on ONE device: what each of 8 GPUs would share out; the 8-GPU run is the driver's
        try:
            out["other_configs"]["C5_image_on_one_gpu"] = other_config_line(sg, fr, "C5's image, one GPU", "default", 1024, "f64", 2, 1,
                                                                            edge=65536)
            out["other_configs"]["C5_image_on_one_gpu"].pop("_compare", None)
        except Exception as e:  # noqa: BLE001  (a smaller card: not an error of the headline)
            out["other_configs"]["C5_image_on_one_gpu"] = {"error": repr(e)}
        sg.image = None
        torch.cuda.empty_cache()
        # Algo::BarnsleyFern, Config::new(fern)'s own size and point count, into a host buffer
        fcfg = fr.Config.new(fr.Algo.BarnsleyFern)
        fr.get_image_fern(fcfg, 1, 1)
        tf = time.perf_counter()
        for k in range(5):
            fr.get_image_fern(fcfg, 1, 2 + k)
        fern_ms = (time.perf_counter() - tf) / 5 * 1e3
        out["other_configs"]["fern"] = {
            "workload": "barnsley fern %dx%d, %d points, threads=1 (Config::new(Algo::BarnsleyFern)), host buffer" % (
                fcfg.width, fcfg.height, fcfg.iterations),
            "ms_per_call": fern_ms, "value": fcfg.iterations / (fern_ms * 1e-3), "unit": "points/s",
            "note": "chaos game on the GPU (fr_render_fern_rgb8); not part of the headline metric"}
        img = sg.render(cfg, prec)  # the shared device buffer held the other configs meanwhile
        torch.cuda.synchronize(device)

    if not args.no_cpu_baseline:
        cores = usable_cores()
        threads = args.cpu_threads or cores
        # the whole image when that is ~10-30 s of CPU work, else a regular sub-sample of it
        est_rate = 5.5e8 * threads
        sample = 1
        while total / (sample * sample) / est_rate > 30.0:
            sample *= 2
        info, colours, cpu_total = cpu_baseline(bytes(cfg), int(prec), threads, sample)
        got = img[::sample, ::sample].cpu().numpy()
        gpu_total, _ = fr.count_iterations(cfg, 0, cfg.height, sample, sample, prec)
        base = {
            "value": info["value"],
            "unit": "pixel-iterations/s",
            "cores": threads,
            "kind": "port",
            "sample": ("the whole image" if sample == 1 else "every %d-th pixel in x and y of the same image" % sample)
                      + " (%d pixels, %d pixel-iterations, %.2f s); oracle/fractal_oracle.c row-parallel driver "
                        "(dynamic row scheduling, like the reference's rayon loop), -O2 -ffp-contract=off, libm log2"
                      % (info["pixels"], info["pixel_iterations"], info["seconds"]),
            "cpu_model": cpu_model(),
            "host_cores_online": os.cpu_count(),
            "cores_usable_by_this_process": cores,
            "gpu_bytes_identical_on_sample": bool((got == colours).all()),
            "gpu_iteration_sum_identical_on_sample": bool(gpu_total == cpu_total),
        }
        if threads > 16:
            i16, _, _ = cpu_baseline(bytes(cfg), int(prec), 16, max(sample, 2))
            base["value_16_threads"] = i16["value"]
        if is_c2:
            # BASELINE C1 — the README's frame (README.md:9-11, examples.md:29: 3000x3000 -s 1e6 -i 1024), "the repo's existing
            # rayon CPU path": the CPU leg in full ("C1 and C2 timed in full") beside the drop-in call for it
            import numpy as np

            c1 = make_config(fr, "zoom1e6", 3000, 1024)
            i1, colours1, total1 = cpu_baseline(bytes(c1), int(fr.Precision.F64), threads, 1)
            buf1 = np.empty((3000, 3000, 3), dtype=np.uint8)
            fr.get_image_rows(c1, 0, 3000, fr.Precision.F64, out=buf1)
            calls = []
            for _ in range(10):
                tc = time.perf_counter()
                fr.get_image_rows(c1, 0, 3000, fr.Precision.F64, out=buf1)
                calls.append((time.perf_counter() - tc) * 1e3)
            m1 = sg.measure(c1, fr.Precision.F64, 10, 2)
            base["C1_frame"] = {
                "workload": "mandelbrot 3000x3000 max_iter=1024 f64 view=zoom1e6 (BASELINE C1, the README's frame), whole frame",
                "cpu_seconds": i1["seconds"], "cpu_value": i1["value"], "cpu_threads": threads, "pixel_iterations": total1,
                "gpu_kernel_ms": m1["kernel_ms"], "gpu_kernel": m1["kernel"], "gpu_value_kernel_only": total1 / (m1["kernel_ms"] * 1e-3),
                "gpu_host_call_ms_median": sorted(calls)[5],
                "gpu_iteration_sum_identical": bool(m1["total"] == total1),
                "gpu_bytes_identical": bool((buf1 == colours1).all()),
                "note": "get_image for the README's frame: the oracle's row-parallel driver on the host cores against "
                        "fr_render_rows_rgb8 into a host buffer (kernel + D2H); the reference quotes '~1 second' for it on a laptop (README.md:11)"}
        out["cpu_baseline"] = base
        out["gpu_over_cpu"] = rate / info["value"]
    print(json.dumps(out), flush=True)


def workload_edge(args, world):
    if args.size:
        return args.size
    if args.workload == "c5":
        return 65536
    if args.workload == "weak":
        return int(round(16384 * math.sqrt(world)))
    return 16384


def workload_label(args, world, edge):
    if args.view == "default" and args.iterations == 1024 and args.precision == "f64":
        if edge == 16384:
            return "C2's image split over %d GPUs" % world
        if edge == 65536:
            return "C5" if world == 8 else "C5's image over %d GPUs" % world
        if args.workload == "weak":
            return "C2-shaped, weak-scaled"
    return "variant"


def run_in_library(args, torch, fr, lib, native):
    """Plain `python3 bench.py --gpus N`: one process, N GPUs, through fr_init_devices."""
    world = args.gpus
    ndev = torch.cuda.device_count()
    if ndev < world and not args.logical:
        print(json.dumps({
            "metric": "pixel_iterations_per_sec", "value": None, "unit": "pixel-iterations/s", "n_gpus": world,
            "error": "needs %d devices, this box has %d (add --logical for %d logical devices on GPU 0: a plumbing "
                     "check, not a scaling number)" % (world, ndev, world),
            "devices_visible": ndev}), flush=True)
        return
    devices = [0] * world if ndev < world else list(range(world))
    logical = ndev < world
    import numpy as np

    torch.cuda.set_device(devices[0])
    fr.init(devices[0])
    fr.init_devices(devices)
    native.check(lib.fr_set_multi_root_share(args.root_share))
    prec = fr.Precision.F32 if args.precision == "f32" else fr.Precision.F64
    edge = workload_edge(args, world)
    cfg = make_config(fr, args.view, edge, args.iterations)
    need = 3 * cfg.width * cfg.height
    gather = native.FR_GATHER_PEER_COPY if (args.gather == "peer" or logical) else native.FR_GATHER_RCCL
    d_img = torch.empty(need, dtype=torch.uint8, device=torch.device("cuda", devices[0]))

    def step():
        native.check(lib.fr_render_rgb8_multi_device(C.byref(cfg), int(prec), args.block_rows, gather, d_img.data_ptr(), need))

    gather_note = None
    if gather == native.FR_GATHER_RCCL:
        try:
            step()
        except native.FractalHipError as e:  # no usable RCCL in this process: the peer-DMA gather does the same job
            gather_note = "RCCL gather unavailable (%s); fell back to peer-to-peer DMA" % e
            gather = native.FR_GATHER_PEER_COPY
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()  # returns when the whole image is in the first GPU's HBM
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = fr.multi_stats()
    total, _ = fr.count_iterations(cfg, 0, cfg.height, 1, 1, prec)
    # the host-buffer variant: every GPU DMAs its blocks to the caller's buffer over its own PCIe link
    hbuf = np.empty((cfg.height, cfg.width, 3), dtype=np.uint8)
    fr.get_image_multi(cfg, prec, args.block_rows, out=hbuf)
    th = time.perf_counter()
    fr.get_image_multi(cfg, prec, args.block_rows, out=hbuf)
    host_ms = (time.perf_counter() - th) * 1e3
    same = bool((hbuf[::61] == d_img.view(cfg.height, cfg.width, 3)[::61].cpu().numpy()).all())
    kmax = max(st["kernel_ms"])
    pixels = cfg.width * cfg.height
    out = {
        "metric": "pixel_iterations_per_sec", "value": total * args.steps / dt, "unit": "pixel-iterations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak" if args.workload == "weak" else "strong", "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic (deterministic: the image is a pure function of the Config)",
        "config": {
            "workload": "%s %dx%d max_iter=%d %s view=%s (BASELINE %s)" % (
                VIEWS[args.view][0], cfg.width, cfg.height, cfg.iterations, args.precision, args.view,
                workload_label(args, world, edge)),
            "per_gpu_pixels": pixels // world,
            "partition": "row-block-cyclic, %d-row blocks, %d devices, ONE process (fr_init_devices: one host thread "
                         "+ streams per device)" % (args.block_rows, world),
            "exchange": ("grouped ncclSend/ncclRecv (ncclCommInitAll)" if gather == native.FR_GATHER_RCCL else
                         "peer-to-peer DMA (hipMemcpyPeerAsync)") + " of finished row blocks to the first device, "
                        "pipelined behind the rendering, inside the timed step",
            "logical_devices_on_one_gpu": logical,
        },
        "mpixels_per_sec": pixels * args.steps / dt / 1e6,
        "pixel_iterations_per_image": total,
        "per_device_kernel_ms": st["kernel_ms"],
        "root_share": args.root_share,
        "per_device": {
            "devices": [{"device": d, "rows": st["rows"][d], "launches": st["kernels"][d], "kernel_ms": st["kernel_ms"][d],
                         "transfer_span_ms": st["transfer_span_ms"][d], "job_ms": st["job_ms"][d],
                         "idle_ms": max(st["job_ms"][d] - st["kernel_ms"][d], 0.0), "bytes_moved": st["bytes_moved"][d],
                         "link_GBps": st["bytes_moved"][d] / (st["transfer_span_ms"][d] * 1e-3) / 1e9 if st["transfer_span_ms"][d] > 0 else None}
                        for d in range(world)],
            "call_wall_ms": st["wall_ms"],
            "note": "the LAST timed step (fr_multi_last_stats), ms: kernel_ms = the device's chunk kernels summed (HIP events); "
                    "transfer_span_ms = device time from its first transfer being ready to its last one done; job_ms = host wall "
                    "time of the device's thread, first launch to streams drained; bytes_moved = what its link to the first "
                    "device carried",
            "prediction": multi_gpu_prediction(world, pixels, total),
        },
        "build_id": fr.build_id(),
        "roofline": dict(roofline_block(args.precision, total / world, kmax, pixels // world,
                                        "escape kernels of the slowest device's share (summed over its chunk launches)",
                                        None), note=ROOFLINE_NOTE),
        "host_buffer_variant": {"ms": host_ms, "value": total / (host_ms * 1e-3), "bytes_identical_to_gathered_image": same,
                                "note": "fr_render_rgb8_multi: each device DMAs its blocks straight to their place in the "
                                        "caller's pinned host buffer over its own PCIe link; resident buffer"},
    }
    if gather_note:
        out["config"]["exchange_note"] = gather_note
    if logical:
        out["note"] = "logical devices share ONE GPU: this line checks the plumbing and is not a scaling measurement"
    print(json.dumps(out), flush=True)


def multi_gpu_prediction(world, pixels, total_px_it):
    """DESIGN.md §4's prediction for this workload at `world` GPUs, next to the measured per-rank fields, so that the first
    real run can be read against it.  Inputs: the single-GPU rate of the default bench line (5.2e12 px-it/s, C2) and
    ~50 GB/s one way over ONE xGMI link (MI355X_MICROARCH.md: 7 links x ~153 GB/s raw per GPU; a single peer-to-root
    stream sustains about a third of a link's raw rate).  A PREDICTION: nothing here was measured on more than one GPU."""
    if world <= 1:
        return None
    render_ms = total_px_it / 5.2e12 / world * 1e3
    bytes_per_peer = 3.0 * pixels / world
    link_ms = bytes_per_peer / 50e9 * 1e3
    step_ms = max(render_ms, link_ms) + 0.25 * min(render_ms, link_ms)  # pipelined in chunks: the shorter leg mostly hides
    ideal = total_px_it / 5.2e12 * 1e3 / world
    return {"render_ms_per_rank": render_ms, "bytes_per_peer": bytes_per_peer, "link_ms_per_peer_at_50GBps": link_ms,
            "predicted_step_ms": step_ms, "predicted_scaling_efficiency": ideal / step_ms,
            "bound": "gather (each peer's link)" if link_ms > render_ms else "render",
            "note": "prediction from single-GPU measurements (DESIGN.md 4); compare per_rank.ranks[*].transfer_span_ms with "
                    "link_ms_per_peer and kernel_ms with render_ms_per_rank; if rank 0's step is the longest, try --root-share 2"}


def run_distributed(args, torch, fr, lib, native, world, rank, local_rank):
    """Under a launcher: one process per GPU, torch.distributed point-to-point (= RCCL) gather."""
    import torch.distributed as dist

    from fractal_renderer_amd import partition as P

    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    fr.init(local_rank)
    prec = fr.Precision.F32 if args.precision == "f32" else fr.Precision.F64
    edge = workload_edge(args, world)
    cfg = make_config(fr, args.view, edge, args.iterations)
    row_bytes = 3 * cfg.width
    B = args.block_rows
    renderer = P.DistributedRenderer(cfg, prec, B, device=device, root_share=args.root_share)
    stream = torch.cuda.current_stream(device)

    def fence():
        torch.cuda.synchronize(device)
        dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        renderer.render()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        renderer.render()
    fence()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    # the dominant kernel's duration for the roofline: this rank's whole share as ONE launch, timed
    # with HIP events outside the timed region (the step itself launches per chunk of blocks)
    # one more step, outside the timed region, with device events around every chunk kernel and every step's transfers:
    # where each rank's time goes (the first real N > 1 run should be diagnostic)
    renderer.render(diagnose=True)
    fence()
    report = renderer.step_report()
    reports = [None] * world
    dist.all_gather_object(reports, report)
    rows_mine = P.local_rows(cfg.height, B, rank, world, args.root_share)
    share = torch.empty(max(rows_mine * row_bytes, 1), dtype=torch.uint8, device=device)
    native.check(lib.fr_set_profiling(1))
    kernel_ms = []
    name = C.create_string_buffer(160)
    name.value = b"(this rank renders nothing: --root-share 0)"
    for _ in range(3):
        tot = 0.0
        for first, stride in P.shares(rank, world, args.root_share):
            rows = C.c_uint64(0)
            native.check(lib.fr_render_block_cyclic_rgb8_device(C.byref(cfg), int(prec), B, first, stride, share.data_ptr(), share.numel(),
                                                                stream.cuda_stream, C.byref(rows)))
            if rows.value:
                ms = C.c_float(0)
                native.check(lib.fr_last_kernel_ms(C.byref(ms)))
                tot += ms.value
                native.check(lib.fr_last_kernel_name(name, len(name)))
        kernel_ms.append(tot)
    del share
    y0 = cfg.height * rank // world
    y1 = cfg.height * (rank + 1) // world
    total, _ = fr.count_iterations(cfg, y0, y1, 1, 1, prec)
    t = torch.tensor([total], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    total = int(t.item())
    k = torch.tensor([sum(kernel_ms) / len(kernel_ms)], dtype=torch.float64, device=device)
    dist.all_reduce(k, op=dist.ReduceOp.MAX)
    kavg = float(k.item())
    if rank == 0:
        pixels = cfg.width * cfg.height
        out = {
            "metric": "pixel_iterations_per_sec", "value": total * args.steps / dt, "unit": "pixel-iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak" if args.workload == "weak" else "strong", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic (deterministic: the image is a pure function of the Config)",
            "config": {
                "workload": "%s %dx%d max_iter=%d %s view=%s (BASELINE %s)" % (
                    VIEWS[args.view][0], cfg.width, cfg.height, cfg.iterations, args.precision, args.view,
                    workload_label(args, world, edge)),
                "per_gpu_pixels": pixels // world,
                "partition": "row-block-cyclic, %d-row blocks, %d ranks (one process per GPU)" % (B, world),
                "exchange": "RCCL point-to-point gather of finished row blocks to rank 0, pipelined behind the "
                            "rendering, inside the timed step",
            },
            "mpixels_per_sec": pixels * args.steps / dt / 1e6,
            "pixel_iterations_per_image": total,
            "kernel_ms_avg": kavg,
            "build_id": fr.build_id(),
            "roofline": dict(roofline_block(args.precision, total / world, kavg, pixels // world,
                                            name.value.decode() + " over one rank's whole share (slowest rank)", None),
                             note=ROOFLINE_NOTE),
            "root_share": args.root_share,
            "per_rank": {
                "ranks": reports,
                "note": "ONE extra step after the timed ones, device events on each rank (ms): kernel_ms = its chunk kernels summed; "
                        "compute_span_ms = first kernel start to last kernel end; transfer_ms = its grouped RCCL point-to-point "
                        "calls on the communication stream (sends on a peer, receives on rank 0), summed; step_ms = the step on "
                        "the device; idle_ms = step - compute span.  bytes_sent is what the rank's xGMI link to rank 0 carried.",
                "link_GBps": [r["bytes_sent"] / (r["transfer_span_ms"] * 1e-3) / 1e9 if r and r["transfer_span_ms"] > 0 else None
                              for r in reports],
                "prediction": multi_gpu_prediction(world, pixels, total),
            },
        }
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def supervise(args):
    """Plain `python3 bench.py --gpus N` (N > 1, no launcher).  This process makes NO GPU call (it never imports
    torch): the measured run is a fresh CHILD process with a time limit.  The first real N > 1 run is also the first
    test of the in-library RCCL gather on more than one GPU (VERDICT r02 #2b), so: if the child hangs (killed with its
    whole process group at the limit) or exits non-zero, a second fresh child runs the same workload with the
    peer-to-peer DMA gather; if that fails too, a JSON line with an "error" field is printed and the exit code is 1.
    A process that has touched the GPU is never re-executed."""
    import signal
    import subprocess

    base = [sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--child"] + ["--child"]

    def run(extra, limit):
        p = subprocess.Popen(base + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            out, err = p.communicate(timeout=limit)
            return p.returncode, out, err, False
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, signal.SIGKILL)  # the child and whatever it started
            except ProcessLookupError:
                pass
            out, err = p.communicate()
            return -9, out, err, True

    def last_json(out):
        for ln in reversed(out.splitlines()):
            if ln.startswith("{"):
                try:
                    return json.loads(ln)
                except ValueError:
                    pass
        return None

    attempts = []
    for extra, what in (([], "gather as requested (--gather %s)" % args.gather), (["--gather", "peer"], "peer-to-peer DMA gather")):
        if attempts and args.gather == "peer":
            break  # the request WAS the peer gather: nothing else to fall back on
        limit = args.child_timeout
        if os.environ.get("FR_BENCH_TEST_STALL") == (extra[1] if extra else args.gather):
            # test hook: only the child that is TOLD to stall gets the short limit, so a healthy child whose first
            # `import torch` on a cold box takes a minute is not mistaken for a hung one
            limit = float(os.environ.get("FR_BENCH_TEST_STALL_TIMEOUT", limit))
        rc, out, err, timed_out = run(extra, limit)
        d = last_json(out)
        if not attempts:
            limit0 = limit
        attempts.append({"what": what, "exit_code": rc, "timed_out": timed_out,
                         "stderr_tail": err[-600:] if rc != 0 else ""})
        if rc == 0 and d is not None:
            if len(attempts) > 1:
                d["fallback"] = {"reason": "the first child %s" % ("hung and was killed after %.0f s" % limit0
                                                                     if attempts[0]["timed_out"] else
                                                                     "exited with code %d" % attempts[0]["exit_code"]),
                                 "attempts": attempts}
            print(json.dumps(d), flush=True)
            return 0
    print(json.dumps({"metric": "pixel_iterations_per_sec", "value": None, "unit": "pixel-iterations/s", "n_gpus": args.gpus,
                      "error": "every measured child process failed", "attempts": attempts}), flush=True)
    return 1


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # FR_BENCH_DISTRIBUTED=1: take the one-process-per-GPU path even at world size 1 (a one-GPU box then exercises
    # its process-group set-up, collectives and bookkeeping; tests/test_gpu_multi.py)
    launched = "RANK" in os.environ and (world > 1 or os.environ.get("FR_BENCH_DISTRIBUTED") == "1")
    if args.gpus > 1 and not launched and not args.child:
        sys.exit(supervise(args))
    if args.child and os.environ.get("FR_BENCH_TEST_STALL") == args.gather:
        time.sleep(1e6)  # test hook: a child that never comes back (tests/test_abi_cpu.py kills it through the supervisor)

    import torch  # first: the library then shares torch's HIP runtime

    import fractal_renderer_amd as fr
    from fractal_renderer_amd import _native

    if not torch.cuda.is_available():
        if args.gpus > 1:
            print(json.dumps({"metric": "pixel_iterations_per_sec", "value": None, "n_gpus": args.gpus,
                              "error": "needs %d devices, this box has 0 (there is no CPU fallback)" % args.gpus}), flush=True)
            return
        sys.exit("bench.py needs a HIP device (there is no CPU fallback)")
    lib = _native.load()
    _native.check(lib.fr_set_tile(args.tile))
    _native.check(lib.fr_set_loop_mode(args.loop_mode))
    _native.check(lib.fr_set_colour_filter(0 if args.no_colour_filter else args.colour_filter))
    if args.cycle_shortcut:
        _native.check(lib.fr_set_cycle_shortcut(1))
    if args.refill:
        mr, q16 = (int(v) for v in args.refill.split(","))
        _native.check(lib.fr_set_refill_policy(mr, q16))
    if launched:
        args.gpus = world
        try:
            run_distributed(args, torch, fr, lib, _native, world, rank, local_rank)
        except Exception as e:  # noqa: BLE001  (leave a parsable line behind, then fail loudly)
            if rank == 0:
                print(json.dumps({"metric": "pixel_iterations_per_sec", "value": None, "unit": "pixel-iterations/s",
                                  "n_gpus": world, "error": "%s: %s" % (type(e).__name__, e)}), flush=True)
            raise
    elif args.gpus > 1:
        run_in_library(args, torch, fr, lib, _native)
    else:
        run_single(args, torch, fr, lib, _native)


if __name__ == "__main__":
    main()
