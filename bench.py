#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on BASELINE.json's config, on N MI355X of one node.

Metric: pixel-iterations/s: Σ over pixels of EXECUTED loop iterations ÷ time, the
Σ being an exact integer counted on the device outside the timed region (and cross-checked against
the CPU oracle on the sampled pixels of the cpu_baseline leg).

Workload at N GPUs (weak scaling, per-GPU pixel count fixed at 16384² = BASELINE config C2):
    Mandelbrot, default view (-x -0.6 -y 0 -s 0.4), max_iter 1024, fp64, image W = H = round(16384·√N)
    (N=1: 16384², N=4: 32768², N=2/8: 23170² / 46341²), row-block-cyclic over the ranks.
A step = one pass of the hot path over the whole image: every rank renders its row blocks into
HBM (inputs are just the Config; outputs stay resident in HBM), and for N > 1 every finished block
is sent to rank 0 over RCCL point-to-point, straight into its place in the full image, while the
next block renders (the path's one real exchange step, north_star: "final RCCL gather over xGMI").

One JSON line on stdout (rank 0).  `roofline` prices the escape+colour kernel against the gfx950
fp64 VECTOR peak (this path is neither HBM- nor MFMA-bound); `cpu_baseline` is the
oracle's row-parallel driver timed on this box's host cores on a bounded sample.
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
VALU_OPS_PER_ITERATION = 8  # after reusing re², im² (bit-safe); FMA is forbidden by bit-exactness


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=16384, help="per-GPU image edge (default: BASELINE C2)")
    ap.add_argument("--iterations", type=int, default=1024)
    ap.add_argument("--precision", choices=["f64", "f32"], default="f64")
    ap.add_argument("--view", choices=["default", "zoom1e6", "julia"], default="default")
    ap.add_argument("--block-rows", type=int, default=256)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--refill", default="", help="tuning: minrun,quit16 of the refilling kernel")
    ap.add_argument("--loop-mode", type=int, default=-1, help="tuning: force the orbit loop form (0, 2, 4)")
    ap.add_argument("--cycle-shortcut", action="store_true",
                    help="measure with the exact periodicity shortcut on (never the headline: it skips iterations)")
    ap.add_argument("--force-blocks", action="store_true",
                    help="N=1 only: render block by block like a rank of an N>1 run does (tuning of --block-rows)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = min(16, cores available)")
    return ap.parse_args()


def make_config(fr, args, edge):
    """CLI-default Config (src/lib.rs:34-226: limit 65536, stable_limit 2, exposure 5, inside,
    smooth, default colours) for the chosen view."""
    cfg = fr.Config.new(fr.Algo.Julia if args.view == "julia" else fr.Algo.Mandelbrot)
    cfg.width = cfg.height = edge
    cfg.iterations = args.iterations
    cfg.exposure = 5.0
    if args.view == "default":
        cfg.pos.re, cfg.pos.im = -0.6, 0.0
    elif args.view == "zoom1e6":
        cfg.pos.re, cfg.pos.im = -0.7436447860, 0.1318252536
        cfg.scale.re = cfg.scale.im = 1e6
    else:
        cfg.julia_set.re, cfg.julia_set.im = -0.8, 0.156
    return cfg


def cpu_baseline(cfg_bytes, precision, threads):
    """Time the CPU oracle (test infrastructure, used here ONLY as the reported baseline) on every
    2nd pixel in x and y of the same workload.  Returns (dict, sampled colours, sampled Σ)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    ocfg = O.Config.from_buffer_copy(cfg_bytes)
    sx = sy = 2
    O.set_log2_mode(O.LOG2_SOFT)
    try:
        t0 = time.perf_counter()
        total, npx, colours = O.sample_image(ocfg, sx, sy, precision, threads)
        dt = time.perf_counter() - t0
    finally:
        O.set_log2_mode(O.LOG2_LIBM)
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    info = {
        "value": total / dt,
        "unit": "pixel-iterations/s",
        "cores": threads,
        "kind": "port",
        "sample": "every 2nd pixel in x and y of the same image (%d pixels, %d pixel-iterations, %.2f s); "
                  "oracle/fractal_oracle.c row-parallel driver, -O2 -ffp-contract=off" % (npx, total, dt),
        "cpu_model": model,
        "host_cores_online": os.cpu_count(),
    }
    return info, colours, total, (sx, sy)


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
        args.gpus = world

    import torch  # first: the library then shares torch's HIP runtime
    import torch.distributed as dist

    import fractal_renderer_amd as fr
    from fractal_renderer_amd import _native
    from fractal_renderer_amd import partition as P

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    fr.init(local_rank)
    lib = _native.load()
    _native.check(lib.fr_set_tile(args.tile))
    _native.check(lib.fr_set_loop_mode(args.loop_mode))
    if args.cycle_shortcut:
        _native.check(lib.fr_set_cycle_shortcut(1))
    if args.refill:
        mr, q16 = (int(v) for v in args.refill.split(","))
        _native.check(lib.fr_set_refill_policy(mr, q16))
    device = torch.device("cuda", local_rank)
    prec = fr.Precision.F32 if args.precision == "f32" else fr.Precision.F64

    edge = int(round(args.size * math.sqrt(world)))
    cfg = make_config(fr, args, edge)
    row_bytes = 3 * cfg.width
    B = args.block_rows
    renderer = P.DistributedRenderer(cfg, prec, B, device=device, force_blocks=args.force_blocks)
    stream = torch.cuda.current_stream(device)

    kernel_ms = []

    def step(record):
        # N = 1: one launch renders the whole image in place; N > 1: one launch per owned row block,
        # each block sent to rank 0 while the next one renders (partition.DistributedRenderer)
        img = renderer.render()
        if record and world == 1 and not args.force_blocks:
            ms = C.c_float(0)
            _native.check(lib.fr_last_kernel_ms(C.byref(ms)))  # HIP events on the launch stream
            kernel_ms.append(ms.value)
        return img

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    _native.check(lib.fr_set_profiling(1 if world == 1 else 0))
    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    img = None
    for _ in range(args.steps):
        img = step(True)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # the dominant kernel's duration for the roofline: this rank's whole share as ONE launch,
        # timed with HIP events outside the timed region (the step itself launches per block)
        share = torch.empty(max(P.local_rows(cfg.height, B, rank, world) * row_bytes, 1), dtype=torch.uint8,
                            device=device)
        _native.check(lib.fr_set_profiling(1))
        for _ in range(3):
            P.render_local_hip(cfg, prec, B, rank, world, share, stream.cuda_stream)
            ms = C.c_float(0)
            _native.check(lib.fr_last_kernel_ms(C.byref(ms)))
            kernel_ms.append(ms.value)
        del share

    # Extra, never the headline: the same steps with the exact periodicity shortcut on (bit-identical
    # output, but periodic orbits are fast-forwarded instead of iterated, so it is not a roofline number)
    shortcut = None
    if world == 1 and not args.cycle_shortcut:
        _native.check(lib.fr_set_cycle_shortcut(1))
        _native.check(lib.fr_set_profiling(0))
        ref = img.clone() if img is not None else None
        renderer.render()
        fence()
        ts = time.perf_counter()
        for _ in range(args.steps):
            img2 = renderer.render()
        fence()
        shortcut = {"ms_per_step": (time.perf_counter() - ts) / args.steps * 1e3,
                    "bytes_identical_to_plain_loop": bool(torch.equal(ref, img2))}
        _native.check(lib.fr_set_cycle_shortcut(0))
        _native.check(lib.fr_set_profiling(1))
        img = ref

    # Extra, never the headline: the drop-in call as the reference's caller sees it — fr_render_rows_rgb8
    # into a HOST buffer (kernel + PCIe D2H, pinned + band-pipelined), second call into the same buffer
    host_call = None
    if world == 1 and rank == 0:
        import numpy as np

        hbuf = np.empty((cfg.height, cfg.width, 3), dtype=np.uint8)
        fr.get_image_rows(cfg, 0, cfg.height, prec, out=hbuf)
        th = time.perf_counter()
        fr.get_image_rows(cfg, 0, cfg.height, prec, out=hbuf)
        host_call = {"ms": (time.perf_counter() - th) * 1e3,
                     "note": "fr_render_rows_rgb8 into a resident host buffer: kernel + D2H over PCIe; not `value`"}
        del hbuf

    # exact Σ executed iterations of the whole image, counted on the device outside the timed region
    y0 = cfg.height * rank // world
    y1 = cfg.height * (rank + 1) // world
    total, _ = fr.count_iterations(cfg, y0, y1, 1, 1, prec)
    kavg = sum(kernel_ms) / max(len(kernel_ms), 1)
    my_px_it = None
    if world > 1:
        t = torch.tensor([total], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total = int(t.item())
        k = torch.tensor([kavg], dtype=torch.float64, device=device)
        dist.all_reduce(k, op=dist.ReduceOp.MAX)
        kavg = float(k.item())

    if rank == 0:
        pixels = cfg.width * cfg.height
        rate = total * args.steps / dt
        peak = FP32_VECTOR_PEAK_TFLOPS if args.precision == "f32" else FP64_VECTOR_PEAK_TFLOPS
        # the dominant kernel: per launch it executes this rank's share of the pixel-iterations
        launch_px_it = total / world
        achieved = FLOPS_PER_ITERATION * launch_px_it / (kavg * 1e-3) / 1e12 if kavg > 0 else 0.0
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                rec = json.load(f).get("%dx%d_i%d_%s_%s" % (cfg.width, cfg.height, cfg.iterations, args.precision, args.view))
                if rec:
                    traffic = rec["hbm_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        out = {
            "metric": "pixel_iterations_per_sec",
            "value": rate,
            "unit": "pixel-iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic (deterministic: the image is a pure function of the Config)",
            "config": {
                "workload": "mandelbrot %dx%d max_iter=%d %s view=%s (BASELINE %s)" % (
                    cfg.width, cfg.height, cfg.iterations, args.precision, args.view,
                    "C2" if (world == 1 and args.size == 16384 and args.iterations == 1024 and args.view == "default"
                             and args.precision == "f64") else "C2-shaped, weak-scaled" if args.view == "default" else "variant"),
                "per_gpu_pixels": pixels // world,
                "partition": "row-block-cyclic, %d-row blocks, %d ranks" % (B, world),
                "exchange": "none" if world == 1 else "RCCL point-to-point gather of finished row blocks to rank 0, "
                                                      "pipelined behind the rendering, inside the timed step",
            },
            "mpixels_per_sec": pixels * args.steps / dt / 1e6,
            "pixel_iterations_per_image": total,
            "kernel_ms_avg": kavg,
            "roofline": {
                "bound": "valu_f64" if args.precision == "f64" else "valu_f32",
                "achieved": achieved,
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": achieved / peak,
                "traffic": traffic,
                "kernel": "escape_strip_kernel<%s, RGB, 7 tiles> (fused coordinate map + orbit loop + colour map)"
                          % ("double" if args.precision == "f64" else "float"),
                "algorithmic_flops_per_launch": FLOPS_PER_ITERATION * launch_px_it,
                "frac_of_attainable_no_fma": (VALU_OPS_PER_ITERATION * launch_px_it / (kavg * 1e-3)) / (peak / 2 * 1e12)
                if kavg > 0 else 0.0,
                # the loop's own share of the SIMDs' issue slots at the nominal clock: 6.5 VALU instructions
                # per iteration (scaled loop, escape check every 4th), each 4 cycles per wave64 for f64
                # (measured: tools/ubench/valu_rates.hip), 1024 SIMDs at 2.4 GHz
                "loop_valu_issue_frac": (6.5 * launch_px_it / 64 * (4 if args.precision == "f64" else 2))
                / (kavg * 1e-3 * 2.4e9 * 1024) if kavg > 0 else 0.0,
                "hbm_check": {"algorithmic_bytes_per_launch": 3 * pixels // world,
                              "achieved_GBps": 3 * pixels / world / (kavg * 1e-3) / 1e9 if kavg > 0 else 0.0,
                              "peak_GBps": 8000.0},
                "note": "bound = fp64 VECTOR issue rate (no MFMA; not HBM: 3 B/pixel written once, see hbm_check). "
                        "frac_of_attainable_no_fma prices SURVEY.md's 8-VALU-op iteration without FMA (peak/2 "
                        "lane-ops/s); the scaled loop issues 6.5 per iteration, so that figure is not a ceiling.",
            },
        }
        if host_call is not None:
            host_call["value"] = total / (host_call["ms"] * 1e-3)
            out["end_to_end_host_buffer"] = host_call
        if shortcut is not None:
            shortcut["value"] = total / (shortcut["ms_per_step"] * 1e-3)
            shortcut["note"] = ("fr_set_cycle_shortcut(1): orbits that return bitwise to an earlier state are "
                                "fast-forwarded to the cap; same bytes, fewer iterations executed; off by default "
                                "and excluded from `value` and `roofline`")
            out["exact_cycle_shortcut"] = shortcut
        if world == 1 and not args.no_cpu_baseline:
            threads = args.cpu_threads or min(16, len(os.sched_getaffinity(0)))
            info, colours, cpu_total, (sx, sy) = cpu_baseline(bytes(cfg), int(prec), threads)
            # byte-compare the GPU image with the CPU path on the sampled pixels, same run
            got = img[::sy, ::sx].cpu().numpy()
            gpu_total, _ = fr.count_iterations(cfg, 0, cfg.height, sx, sy, prec)
            info["gpu_bytes_identical_on_sample"] = bool((got == colours).all())
            info["gpu_iteration_sum_identical_on_sample"] = bool(gpu_total == cpu_total)
            out["cpu_baseline"] = info
            out["gpu_over_cpu"] = rate / info["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
