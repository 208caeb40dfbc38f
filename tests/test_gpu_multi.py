"""get_image across a device SET from one process (fr_init_devices / fr_render_rgb8_multi*), per-call
options, and lifetime safety — through the C ABI, on the GPU box.

A one-GPU box exercises the whole multi-device path with a set that lists its GPU several times
("logical devices": own host thread, streams and scratch each); the results must be byte-identical to
the single render, which test_gpu_parity.py pins against the oracle.
"""
import ctypes as C
import threading

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fr():
    import torch  # noqa: F401  (first: the library then binds to the HIP runtime torch carries)

    import fractal_renderer_amd

    assert fractal_renderer_amd.device_count() > 0, "no HIP device: the GPU tests need a real MI355X"
    fractal_renderer_amd.init(0)
    return fractal_renderer_amd


@pytest.fixture(scope="module")
def lib(fr):
    from fractal_renderer_amd import _native

    return _native.load()


def cfg_of(fr, width, height, iterations=200, algo=0, **kw):
    ocfg = O.cli_config(width, height, algo, iterations=iterations, **kw)
    return fr.Config.from_buffer_copy(bytes(ocfg)), ocfg


SHAPES = [
    (2048, 2048, 256, 0),     # C2-shaped (square, default view)
    (1237, 1001, 0, 150),     # ragged: height not a multiple of the block, width not of the tile
    (513, 77, 8, 64),         # fewer blocks than devices for the larger block sizes
    (64, 2500, 64, 40),       # tall and thin
]


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0], [0] * 5])
def test_multi_host_buffer_matches_single_render(fr, devices):
    fr.init_devices(devices)
    for width, height, block_rows, iters in SHAPES:
        cfg, _ = cfg_of(fr, width, height, iters or 200)
        want = fr.get_image(cfg)
        got = fr.get_image_multi(cfg, 0, block_rows)
        assert np.array_equal(got, want), (devices, width, height, block_rows)
        st = fr.multi_stats()
        assert st["n_devices"] == len(devices) and sum(st["rows"]) == height
        assert all(k > 0 for k, r in zip(st["kernels"], st["rows"]) if r > 0)


def test_multi_matches_oracle_directly(fr):
    """Not only equal to the single render: equal to the CPU oracle (libm log2, what the reference calls)."""
    fr.init_devices([0, 0, 0])
    cfg, ocfg = cfg_of(fr, 777, 333, 300)
    assert np.array_equal(fr.get_image_multi(cfg, 0, 16), O.get_image(ocfg))
    cfg, ocfg = cfg_of(fr, 640, 400, 500, algo=O.JULIA, julia_set=(-0.8, 0.156))
    assert np.array_equal(fr.get_image_multi(cfg, 1, 64), O.get_image(ocfg, O.F32))


@pytest.mark.parametrize("gather", [0, 1])
@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
def test_multi_gather_into_device_memory(fr, lib, devices, gather):
    import torch
    from fractal_renderer_amd import _native

    fr.init_devices(devices)
    for width, height, block_rows, iters in SHAPES[:3]:
        cfg, _ = cfg_of(fr, width, height, iters or 200)
        want = fr.get_image(cfg)
        d_out = torch.zeros(height * width * 3, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()  # the fill is on torch's stream, the render on the library's own
        rc = lib.fr_render_rgb8_multi_device(C.byref(cfg), 0, block_rows, gather, d_out.data_ptr(), d_out.numel())
        if gather == _native.FR_GATHER_RCCL and len(devices) > 1:
            # a communicator cannot hold one GPU twice: the call must say so, not hang
            assert rc == _native.FR_ERR_INVALID_ARGUMENT and b"distinct devices" in lib.fr_last_error()
            continue
        _native.check(rc)
        assert np.array_equal(d_out.cpu().numpy().reshape(height, width, 3), want), (devices, gather, width, height)


def test_rccl_selftest_on_one_device(fr, lib):
    """The only RCCL traffic a one-GPU box can carry: librccl loads, ncclCommInitAll works, a grouped
    self send/recv delivers the bytes."""
    from fractal_renderer_amd import _native

    fr.init_devices([0])
    _native.check(lib.fr_debug_rccl_selftest(1 << 20))
    _native.check(lib.fr_debug_rccl_selftest(3 * 1237 * 8))


def test_multi_full_size_c2_shape_three_logical_devices(fr):
    """BASELINE C2's image (16384^2, 805 MB) over three logical devices into a fresh host buffer."""
    fr.init_devices([0, 0, 0])
    cfg, ocfg = cfg_of(fr, 16384, 16384, 1024)
    got = fr.get_image_multi(cfg)
    want = fr.get_image(cfg)
    assert np.array_equal(got, want)
    # and a sample of it against the oracle in libm mode (what the reference computes)
    total, npx, colours = O.sample_image(ocfg, 64, 64, O.F64, 0)
    assert np.array_equal(got[::64, ::64], colours)


def test_multi_large_julia_takes_two_passes_per_device_chunk(fr, lib):
    """A Julia image whose per-device chunks are large enough (>= 65 536 tiles) for the default dispatch to render
    them in two passes: survivor lists per chunk, results written at their image rows; host buffer and device gather."""
    import torch
    from fractal_renderer_amd import _native

    fr.init_devices([0, 0])
    cfg, ocfg = cfg_of(fr, 16384, 8192, 300, algo=O.JULIA, julia_set=(-0.8, 0.156))
    got = fr.get_image_multi(cfg)
    total, npx, colours = O.sample_image(ocfg, 32, 32, O.F64, 0)
    assert np.array_equal(got[::32, ::32], colours)
    single = torch.empty(8192 * 16384 * 3, dtype=torch.uint8, device="cuda:0")
    s = torch.cuda.current_stream()
    o = fr.RenderOpts(tile=8)  # the strip kernel
    _native.check(lib.fr_render_rows_rgb8_device_opts(C.byref(cfg), 0, 0, 8192, single.data_ptr(), single.numel(), s.cuda_stream,
                                                      C.byref(o)))
    torch.cuda.synchronize()
    assert torch.equal(torch.from_numpy(got).reshape(-1), single.cpu())
    d_out = torch.zeros(8192 * 16384 * 3, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    _native.check(lib.fr_render_rgb8_multi_device(C.byref(cfg), 0, 0, _native.FR_GATHER_PEER_COPY, d_out.data_ptr(), d_out.numel()))
    assert torch.equal(d_out, single)
    st = fr.multi_stats()
    assert st["n_devices"] == 2 and sum(st["rows"]) == 8192


def test_multi_argument_errors(fr, lib):
    from fractal_renderer_amd import _native

    fr.init_devices([0, 0])
    cfg, _ = cfg_of(fr, 100, 60)
    out = np.empty((60, 100, 3), dtype=np.uint8)
    assert lib.fr_render_rgb8_multi(C.byref(cfg), 0, 12, out.ctypes.data, out.nbytes) == _native.FR_ERR_INVALID_ARGUMENT
    assert lib.fr_render_rgb8_multi(C.byref(cfg), 0, 8, out.ctypes.data, 10) == _native.FR_ERR_BUFFER_TOO_SMALL
    assert lib.fr_render_rgb8_multi(C.byref(cfg), 7, 8, out.ctypes.data, out.nbytes) == _native.FR_ERR_INVALID_ARGUMENT
    assert lib.fr_render_rgb8_multi(None, 0, 8, out.ctypes.data, out.nbytes) == _native.FR_ERR_INVALID_ARGUMENT
    bad = (C.c_int * 2)(0, 99)
    assert lib.fr_init_devices(bad, 2) == _native.FR_ERR_NO_DEVICE
    assert lib.fr_init_devices(bad, 0) == _native.FR_ERR_INVALID_ARGUMENT
    # the failed init left the old set in place
    n = C.c_int(0)
    _native.check(lib.fr_multi_device_count(C.byref(n)))
    assert n.value == 2
    fr.shutdown()
    assert lib.fr_render_rgb8_multi(C.byref(cfg), 0, 8, out.ctypes.data, out.nbytes) == _native.FR_ERR_INVALID_ARGUMENT
    assert b"fr_init_devices" in lib.fr_last_error()
    fr.init(0)


def test_multi_empty_image_and_fern(fr):
    fr.init_devices([0, 0])
    cfg, _ = cfg_of(fr, 0, 10)
    assert fr.get_image_multi(cfg).size == 0
    cfg, _ = cfg_of(fr, 40, 30, algo=O.BARNSLEY_FERN)
    assert not fr.get_image_multi(cfg, 0, 8).any()  # calc/src/lib.rs:211: BLACK on this path


# ---- per-call options ----------------------------------------------------------------------------


def test_per_call_opts_are_independent_of_the_process_defaults(fr, lib):
    """Two threads render concurrently with different selectors; every combination gives the single
    render's bytes, and the process-wide defaults are untouched."""
    from fractal_renderer_amd import _native

    cfg, ocfg = cfg_of(fr, 700, 520, 400)
    jcfg, jocfg = cfg_of(fr, 700, 520, 400, algo=O.JULIA, julia_set=(-0.8, 0.156))
    want = {0: O.get_image(ocfg), 2: O.get_image(jocfg)}
    combos = [dict(tile=9, loop_mode=0), dict(tile=8, cycle_shortcut=1, colour_filter=0), dict(tile=1, loop_mode=2),
              dict(tile=9, cycle_shortcut=1, refill_minrun=4, refill_quit16=3), dict(tile=808, palette=0),
              dict(tile=10), dict(tile=10, cycle_shortcut=1, colour_filter=0)]
    errors = []

    def worker(k):
        try:
            for rep in range(6):
                kw = combos[(k + rep) % len(combos)]
                c = jcfg if (k + rep) % 2 else cfg
                got = fr.get_image_rows(c, 0, c.height, 0, opts=fr.RenderOpts(**kw))
                if not np.array_equal(got, want[int(c.algo)]):
                    errors.append((k, rep, kw))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    d = fr.RenderOpts()
    assert (d.tile, d.loop_mode, d.palette, d.cycle_shortcut, d.colour_filter) == (0, -1, 1, 0, 1)
    bad = fr.RenderOpts()
    bad.tile = 3
    out = np.empty((cfg.height, cfg.width, 3), dtype=np.uint8)
    assert lib.fr_render_rows_rgb8_opts(C.byref(cfg), 0, 0, cfg.height, out.ctypes.data, out.nbytes,
                                        C.byref(bad)) == _native.FR_ERR_INVALID_ARGUMENT
    bad = fr.RenderOpts()
    bad.size = 4
    assert lib.fr_render_rows_rgb8_opts(C.byref(cfg), 0, 0, cfg.height, out.ctypes.data, out.nbytes,
                                        C.byref(bad)) == _native.FR_ERR_INVALID_ARGUMENT


def test_fr_init_and_shutdown_while_device_pointer_calls_are_in_flight(fr, lib):
    """fr_init / fr_shutdown take the lifetime lock exclusively: state is never torn down under a call
    that is using it (threads keep rendering through the device-pointer API meanwhile)."""
    import torch
    from fractal_renderer_amd import _native

    cfg, ocfg = cfg_of(fr, 512, 384, 300, smooth=0)  # smooth off: uses the palette slot ring
    want = O.get_image(ocfg)
    stop = threading.Event()
    errors = []

    def renderer():
        try:
            torch.cuda.set_device(0)
            s = torch.cuda.Stream()
            d = torch.empty(want.size, dtype=torch.uint8, device="cuda:0")
            while not stop.is_set():
                _native.check(lib.fr_render_rows_rgb8_device(C.byref(cfg), 0, 0, cfg.height, d.data_ptr(), d.numel(),
                                                             s.cuda_stream))
                s.synchronize()
                if not np.array_equal(d.cpu().numpy().reshape(want.shape), want):
                    errors.append("wrong bytes")
                    return
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=renderer) for _ in range(3)]
    for t in threads:
        t.start()
    for k in range(40):
        if k % 4 == 3:
            fr.shutdown()
        fr.init(0 if k % 2 else -1)
    stop.set()
    for t in threads:
        t.join()
    assert not errors, errors


def test_last_kernel_name_reports_what_ran(fr, lib):
    import torch
    from fractal_renderer_amd import _native

    buf = C.create_string_buffer(128)
    d = torch.empty(2048 * 2048 * 3, dtype=torch.uint8, device="cuda:0")
    s = torch.cuda.current_stream()
    _native.check(lib.fr_set_profiling(1))
    try:
        # (4096^2: the default dispatch samples the view and picks among the strip kernel and the first-pass kernel, with or
        # without lists — whichever it is, the name reported is the kernel that ran)
        for algo, expect in [(0, (b"escape_strip_kernel<double", b"escape_first_kernel")), (O.JULIA, (b"escape_strip_kernel<double", b"escape_first_kernel"))]:
            cfg, _ = cfg_of(fr, 4096, 4096, 100, algo=algo, julia_set=(-0.8, 0.156))
            d = torch.empty(4096 * 4096 * 3, dtype=torch.uint8, device="cuda:0")
            _native.check(lib.fr_render_rows_rgb8_device(C.byref(cfg), 0, 0, 4096, d.data_ptr(), d.numel(), s.cuda_stream))
            _native.check(lib.fr_last_kernel_name(buf, len(buf)))
            assert buf.value.startswith(expect), buf.value
            ms = C.c_float(0)
            _native.check(lib.fr_last_kernel_ms(C.byref(ms)))
            assert ms.value > 0
    finally:
        _native.check(lib.fr_set_profiling(0))


def _bench(args):
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_bench_plain_gpus_n_without_a_launcher():
    """`python3 bench.py --gpus N` from a plain shell: on a box with fewer than N GPUs it prints a JSON line with an
    error field and exits 0; with --logical it drives N logical devices through fr_init_devices, and its two
    variants (gathered in HBM, DMA'd to the host buffer) produce the same bytes."""
    import torch

    if torch.cuda.device_count() < 2:
        d = _bench(["--gpus", "2"])
        assert d["value"] is None and d["n_gpus"] == 2 and "needs 2 devices" in d["error"]
    d = _bench(["--gpus", "3", "--logical", "--size", "3072", "--steps", "2", "--warmup", "1"])
    assert d["n_gpus"] == 3 and d["value"] > 1e9 and d["scaling"] == "strong"
    assert d["config"]["logical_devices_on_one_gpu"] is True and "ONE process" in d["config"]["partition"]
    assert len(d["per_device_kernel_ms"]) == 3 and all(ms > 0 for ms in d["per_device_kernel_ms"])
    assert d["host_buffer_variant"]["bytes_identical_to_gathered_image"] is True
    assert 0 < d["roofline"]["frac"] < 1


def test_bench_under_the_launcher_at_world_size_one():
    """The driver's own invocation form — python -m torch.distributed.run ... bench.py --gpus N — with N = 1 and the
    one-process-per-GPU path forced: RCCL process group on the GPU, barriers, all-reduces, the per-rank roofline
    launch, the JSON line.  (N > 1 needs N GPUs: the driver's to run.)"""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FR_BENCH_DISTRIBUTED="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--size", "4096"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 1e9 and "one process per GPU" in d["config"]["partition"]
    assert 0 < d["roofline"]["frac"] < 1 and d["roofline"]["kernel"].startswith("escape_strip_kernel<double")


@pytest.mark.parametrize("sink", ["host", "peer"])
def test_multi_device_failure_drains_and_recovers(fr, lib, sink):
    """a failure inside device_job's chunk loop used to return straight out of the function with
    kernels and DMAs in flight into memory the caller frees next.  Injected here on one logical device of three,
    at its first and at a later chunk: the call must come back with an error (no hang), and the render after it
    — same set, same buffers — must be byte-identical to the single render."""
    import torch

    fr.init_devices([0, 0, 0])
    cfg, _ = cfg_of(fr, 1536, 4096, 180)
    want = fr.get_image(cfg)
    for dev, chunk in ((1, 0), (2, 1), (0, 2)):
        from fractal_renderer_amd import _native

        _native.check(lib.fr_debug_inject_multi_failure(dev, chunk))
        out = np.zeros((cfg.height, cfg.width, 3), dtype=np.uint8)
        if sink == "host":
            rc = lib.fr_render_rgb8_multi(C.byref(cfg), 0, 64, out.ctypes.data, out.nbytes)
        else:
            d = torch.zeros(out.nbytes, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            rc = lib.fr_render_rgb8_multi_device(C.byref(cfg), 0, 64, 0, C.c_void_p(d.data_ptr()), d.numel())
        assert rc == 4, (dev, chunk, rc)  # FR_ERR_HIP
        msg = lib.fr_last_error().decode()
        assert "device %d" % dev in msg and "injected failure" in msg, msg
        _native.check(lib.fr_debug_inject_multi_failure(-1, 0))
        got = fr.get_image_multi(cfg, 0, 64)
        assert np.array_equal(got, want), (dev, chunk)
    fr.init_devices([0])


def test_multi_device_failure_rccl_sink_needs_two_gpus(fr, lib):
    """the same injected failure with sink = RCCL.  Each rank aborts only ITS OWN communicator, from its owner
    thread (fr_multi.hip: Rccl::abort_own); healthy ranks wait for their transfers by polling and see the abort flag.  A
    communicator cannot hold one GPU twice, so this needs two distinct devices: SKIPPED on the one-GPU box of this
    pipeline — it is here for the first multi-GPU box that runs the suite."""
    import torch

    if fr.device_count() < 2:
        pytest.skip("needs two distinct GPUs (an RCCL communicator cannot hold one GPU twice)")
    from fractal_renderer_amd import _native

    fr.init_devices([0, 1])
    cfg, _ = cfg_of(fr, 1536, 4096, 180)
    want = fr.get_image(cfg)
    d = torch.zeros(want.nbytes, dtype=torch.uint8, device="cuda:0")
    for dev, chunk in ((1, 0), (0, 1), (1, 2)):
        _native.check(lib.fr_debug_inject_multi_failure(dev, chunk))
        rc = lib.fr_render_rgb8_multi_device(C.byref(cfg), 0, 64, 1, C.c_void_p(d.data_ptr()), d.numel())
        assert rc == 4, (dev, chunk, rc)
        assert "injected failure" in lib.fr_last_error().decode()
        _native.check(lib.fr_debug_inject_multi_failure(-1, 0))
        d.zero_()
        torch.cuda.synchronize()
        _native.check(lib.fr_render_rgb8_multi_device(C.byref(cfg), 0, 64, 1, C.c_void_p(d.data_ptr()), d.numel()))
        assert np.array_equal(d.cpu().numpy().reshape(want.shape), want), (dev, chunk)
    fr.init_devices([0])


@pytest.mark.parametrize("q", [2, 4, 0])
def test_the_sink_device_may_render_a_smaller_share(fr, lib, q):
    """fr_set_multi_root_share — the first device keeps a half / a quarter / none of its row blocks, the others
    take them over as extra arithmetic progressions (the dealing of partition.py: shares).  Three and five logical devices,
    host sink and peer gather, ragged and C2-shaped images: the bytes are those of the single render."""
    import torch

    from fractal_renderer_amd import _native

    try:
        _native.check(lib.fr_set_multi_root_share(q))
        for devices in ([0, 0, 0], [0] * 5):
            fr.init_devices(devices)
            for width, height, block_rows, iters in ((1237, 1001, 8, 150), (2048, 4096, 64, 200), (513, 77, 8, 64)):
                cfg, _ = cfg_of(fr, width, height, iters)
                want = fr.get_image(cfg)
                got = fr.get_image_multi(cfg, 0, block_rows)
                assert np.array_equal(got, want), (q, devices, width, height, "host sink")
                st = fr.multi_stats()
                assert sum(st["rows"]) == height
                if q == 0:
                    assert st["rows"][0] == 0 and st["kernels"][0] == 0
                d = torch.zeros(want.nbytes, dtype=torch.uint8, device="cuda:0")
                torch.cuda.synchronize()
                _native.check(lib.fr_render_rgb8_multi_device(C.byref(cfg), 0, block_rows, 0, C.c_void_p(d.data_ptr()), d.numel()))
                assert np.array_equal(d.cpu().numpy().reshape(want.shape), want), (q, devices, width, height, "peer gather")
        assert lib.fr_set_multi_root_share(3) == _native.FR_ERR_INVALID_ARGUMENT
    finally:
        _native.check(lib.fr_set_multi_root_share(1))
        fr.init_devices([0])


def test_multi_host_buffer_pinned_by_the_caller(fr, lib):
    """fr_pin_host_buffer: a frame buffer that is rendered into again and again is pinned once by its owner; the
    multi-device and the single-device host renders find it registered and produce the same bytes."""
    from fractal_renderer_amd import _native

    fr.init_devices([0, 0])
    cfg, _ = cfg_of(fr, 4096, 4096, 120)
    want = fr.get_image(cfg)
    buf = np.zeros((cfg.height, cfg.width, 3), dtype=np.uint8)
    _native.check(lib.fr_pin_host_buffer(C.c_void_p(buf.ctypes.data), buf.nbytes))
    try:
        for _ in range(3):
            buf[:] = 0
            fr.get_image_multi(cfg, 0, 0, out=buf)
            assert np.array_equal(buf, want)
        buf[:] = 0
        _native.check(lib.fr_render_rgb8(C.byref(cfg), buf.ctypes.data, buf.nbytes))
        assert np.array_equal(buf, want)
    finally:
        _native.check(lib.fr_unpin_host_buffer(C.c_void_p(buf.ctypes.data)))
    fr.init_devices([0])
