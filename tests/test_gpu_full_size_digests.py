"""WHOLE-image parity at BASELINE.json's full sizes, at zero oracle cost on the GPU box.

tests/golden/full_size_digests.json holds, for C1, C2 (f64 and f32), C3, C4 (f32 and f64) and C5, the SHA-256 of every
256-row block of what the CPU oracle computes for the WHOLE image — the packed r,g,b bytes get_image returns
(src/lib.rs:253-270), the u32 escape index and the f64 final position `recursive` returns for every pixel
(calc/src/lib.rs:245-257), and the exact executed-iteration sum — generated once in the build container by
tests/golden/make_full_size_digests.py (C3 alone is an hour of its 8 vCPUs).  Here the device's output is hashed block by
block and compared: every pixel of every configuration, the two-pass kernels (C3's and C4's default) included, and the
index ARRAY — not just its sum — at full size.  C5 additionally goes through the 8-way multi-device path BASELINE defines
for it (eight logical devices on the one GPU of this box): host sink and peer gather, block offsets past 2^32.

On a mismatch the first differing block is re-computed with the oracle (seconds to a minute, only then) to say where.
The sampled comparisons of test_gpu_parity.py stay as fast pre-checks.
"""
import ctypes as C
import hashlib
import json
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import golden_util as G
import oracle_lib as O

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "full_size_digests.json")) as _f:
    DIGESTS = json.load(_f)
BLOCK = DIGESTS["block_rows"]
COMPLETE = sorted(n for n, e in DIGESTS["configs"].items() if "executed_total" in e)
POOL = ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1))


@pytest.fixture(scope="module")
def fr():
    import fractal_renderer_amd as fr

    fr.init(0)
    return fr


def sha(a):
    return hashlib.sha256(memoryview(np.ascontiguousarray(a)).cast("B")).hexdigest()


def config_of(fr, name):
    ent = DIGESTS["configs"][name]
    ocfg = G.fill_config(O.Config(), ent["config"])
    cfg = fr.Config.from_buffer_copy(bytes(ocfg))
    prec = fr.Precision.F32 if ent["precision"] == "f32" else fr.Precision.F64
    oprec = O.F32 if ent["precision"] == "f32" else O.F64
    return ent, ocfg, cfg, prec, oprec


def blocks_of(height):
    return [(b, b * BLOCK, min(height, (b + 1) * BLOCK)) for b in range((height + BLOCK - 1) // BLOCK)]


def explain_rgb(name, ocfg, oprec, b, y0, y1, got_rows):
    """First mismatching block: run the oracle on it (only now) and say how many pixels differ and where."""
    want = O.get_image(ocfg, oprec, y0, y1)
    bad = np.argwhere((want != got_rows).any(axis=2))
    return "%s block %d (rows %d..%d): %d pixels differ from the oracle, first at (x, y) %s" % (
        name, b, y0, y1, len(bad), [(int(x), int(y) + y0) for y, x in bad[:5]])


def check_rgb_blocks(name, ocfg, oprec, img, ent):
    assert img.shape == (ocfg.height, ocfg.width, 3)
    bl = blocks_of(ocfg.height)
    got = list(POOL.map(lambda t: sha(img[t[1]:t[2]]), bl))
    assert len(ent["rgb"]) == len(bl)
    for (b, y0, y1), g in zip(bl, got):
        if g != ent["rgb"][b]:
            pytest.fail(explain_rgb(name, ocfg, oprec, b, y0, y1, img[y0:y1]))


@pytest.mark.parametrize("name", COMPLETE)
def test_whole_image_rgb_digests(fr, name):
    """get_image at full size through the default dispatch: every 256-row block's SHA-256 equals the oracle's (libm log2
    mode, which the software log2 reproduced on every block: rgb_soft_differs is empty)."""
    ent, ocfg, cfg, prec, oprec = config_of(fr, name)
    assert ent["rgb_soft_differs"] == {}
    img = fr.get_image(cfg, prec)
    check_rgb_blocks(name, ocfg, oprec, img, ent)
    total, npx = fr.count_iterations(cfg, precision=prec)
    assert (total, npx) == (ent["executed_total"], ocfg.width * ocfg.height)


@pytest.mark.parametrize("name", COMPLETE)
def test_whole_image_escape_index_and_position_digests(fr, name):
    """recursive()'s return value for EVERY pixel at full size (calc/src/lib.rs:245-257): the u32 escape-index array and the
    f64 final positions, hashed per 256-row block, and the per-block executed-iteration sums — not just the image's total."""
    ent, ocfg, cfg, prec, oprec = config_of(fr, name)
    bl = blocks_of(ocfg.height)
    per_call = max(1, (1 << 24) // (ocfg.width * BLOCK))  # ~16 M pixels (0.4 GB of z + indices) per device call
    for k in range(0, len(bl), per_call):
        group = bl[k:k + per_call]
        ya, yb = group[0][1], group[-1][2]
        z, it = fr.escape_rows(cfg, ya, yb, prec)

        def digest(t):
            b, y0, y1 = t
            i = it[y0 - ya:y1 - ya]
            il = i.astype(np.uint64)
            return (sha(i.astype("<u4")), sha(z[y0 - ya:y1 - ya].astype("<f8")),
                    int(np.where(il < ocfg.iterations, il + 1, ocfg.iterations).sum()))

        for (b, y0, y1), (di, dz, ex) in zip(group, POOL.map(digest, group)):
            if di != ent["iters"][b] or dz != ent["z"][b] or ex != ent["executed"][b]:
                wz, wit = O.escape_rows(ocfg, oprec, y0, y1)
                bad_i = np.argwhere(wit != it[y0 - ya:y1 - ya])
                bad_z = np.argwhere((wz.view(np.uint64) != z[y0 - ya:y1 - ya].view(np.uint64)).any(axis=2))
                pytest.fail("%s block %d (rows %d..%d): %d escape indices and %d final positions differ from the oracle, first "
                            "at (y, x) %s / %s" % (name, b, y0, y1, len(bad_i), len(bad_z), bad_i[:3].tolist(), bad_z[:3].tolist()))


def test_c5_through_eight_logical_devices_host_sink_and_peer_gather(fr):
    """BASELINE C5 AS DEFINED: 65536^2 tiled across 8 devices with a gather.  Eight logical devices on this box's one GPU
    render the row blocks cyclically (block b -> device b % 8) — the first time the multi-device path sees an image whose
    in-place block offsets pass 2^32 — (a) into the caller's host buffer, every device DMA-ing its blocks to their final
    place (fr_render_rgb8_multi), and (b) gathered in the first device's HBM by peer copies (fr_render_rgb8_multi_device);
    both compared with the oracle's digests block by block.  (The RCCL gather needs distinct GPUs: the driver's 8-GPU run.)"""
    import torch

    from fractal_renderer_amd import _native

    name = "C5"
    if name not in COMPLETE:
        pytest.skip("no complete C5 digests")
    ent, ocfg, cfg, prec, oprec = config_of(fr, name)
    lib = _native.load()
    fr.init_devices([0] * 8)
    try:
        img = fr.get_image_multi(cfg, int(prec), 0)
        st = fr.multi_stats()
        assert st["n_devices"] == 8 and sum(st["rows"]) == ocfg.height and min(st["rows"]) == ocfg.height // 8
        check_rgb_blocks(name + " (8 logical devices, host sink)", ocfg, oprec, img, ent)
        del img
        nbytes = 3 * ocfg.width * ocfg.height
        d = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        d.zero_()
        torch.cuda.synchronize()  # the fill runs on torch's stream, the render on the library's own: order them
        _native.check(lib.fr_render_rgb8_multi_device(C.byref(cfg), int(prec), 0, _native.FR_GATHER_PEER_COPY,
                                                      C.c_void_p(d.data_ptr()), nbytes))
        torch.cuda.synchronize()
        row_bytes = 3 * ocfg.width
        stage = torch.empty(16 * BLOCK * row_bytes, dtype=torch.uint8, pin_memory=True)
        bl = blocks_of(ocfg.height)
        for k in range(0, len(bl), 16):
            group = bl[k:k + 16]
            a, b = group[0][1] * row_bytes, group[-1][2] * row_bytes
            stage[:b - a].copy_(d[a:b])
            host = stage[:b - a].numpy().reshape(-1, ocfg.width, 3)
            got = list(POOL.map(lambda t: sha(host[t[1] - group[0][1]:t[2] - group[0][1]]), group))
            for (bi, y0, y1), g in zip(group, got):
                if g != ent["rgb"][bi]:
                    pytest.fail(explain_rgb(name + " (8 logical devices, peer gather)", ocfg, oprec, bi, y0, y1,
                                            host[y0 - group[0][1]:y1 - group[0][1]].copy()))
        del d, stage
        torch.cuda.empty_cache()
    finally:
        fr.init_devices([0])


def test_c1_frame_through_the_c_abi_host_path(fr):
    """C1 AS A FRAME (README.md:9-11, examples.md:29: 3000x3000 -s 1e6 -i 1024): fr_render_rgb8 — what the Rust shim's
    get_image calls — into a fresh host buffer; whole image against the digests, f64."""
    from fractal_renderer_amd import _native

    ent, ocfg, cfg, prec, oprec = config_of(fr, "C1")
    assert (cfg.width, cfg.height, cfg.iterations) == (3000, 3000, 1024)
    out = np.empty((3000, 3000, 3), dtype=np.uint8)
    _native.check(_native.load().fr_render_rgb8(C.byref(cfg), out.ctypes.data, out.nbytes))
    check_rgb_blocks("C1 (fr_render_rgb8)", ocfg, oprec, out, ent)
