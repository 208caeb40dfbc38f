"""The multi-GPU path's partition + gather + reassembly, on CPU with gloo (world_size 2 and 3).
The HIP renderer is replaced by a CPU stand-in (the oracle) so only the sharding logic is under
test here; the kernels' own block-cyclic rendering is covered by the -m gpu tests."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_render_rows(ocfg):
    def render(config, precision, y0, y1, out):
        out.view(y1 - y0, config.width, 3).numpy()[:] = O.get_image(ocfg, y0=y0, y1=y1, threads=1)

    return render


def _worker(rank, world, port, w, h, block_rows, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fractal_renderer_amd as fr
        from fractal_renderer_amd import partition as P

        ocfg = O.cli_config(w, h, iterations=60)
        cfg = fr.Config.from_buffer_copy(bytes(ocfg))
        renderer = P.DistributedRenderer(cfg, 0, block_rows, render_rows=_oracle_render_rows(ocfg))
        ok = True
        for _ in range(2):  # buffers are reused across steps
            img = renderer.render()
            if rank == 0:
                ok = ok and bool(np.array_equal(img.numpy(), O.get_image(ocfg, threads=1)))
            else:
                assert img is None
        if rank == 0:
            q.put(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h,block_rows", [(2, 33, 70, 8), (2, 16, 64, 8), (3, 20, 101, 16), (2, 9, 5, 8), (4, 12, 37, 4), (8, 10, 1003, 8)])
def test_block_cyclic_gather_reassembles_the_image(world, w, h, block_rows):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, w, h, block_rows, q)) for r in range(world)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) is True


def test_partition_arithmetic_matches_the_c_abi():
    import __graft_entry__ as ge

    ge.build()
    from fractal_renderer_amd import _native
    from fractal_renderer_amd import partition as P

    f = _native.load().fr_block_cyclic_rows
    for h, b, n in [(100, 8, 3), (16384, 64, 8), (23170, 64, 2), (46341, 64, 8), (5, 8, 2), (64, 64, 4)]:
        rows = [P.local_rows(h, b, r, n) for r in range(n)]
        assert rows == [f(h, b, r, n) for r in range(n)]
        assert sum(rows) == h and rows[0] == max(rows)
        seen = sorted(P.global_row_of(lr, b, r, n) for r in range(n) for lr in range(rows[r]))
        assert seen == list(range(h))


def test_assemble_is_the_inverse_of_the_partition():
    from fractal_renderer_amd import partition as P

    for h, b, n, rb in [(101, 16, 3, 6), (64, 8, 2, 3), (5, 8, 2, 9), (130, 64, 4, 3)]:
        img = torch.arange(h * rb, dtype=torch.int64).remainder(251).to(torch.uint8).view(h, rb)
        max_rows = P.local_rows(h, b, 0, n)
        gathered = torch.zeros((n, max_rows * rb), dtype=torch.uint8)
        for r in range(n):
            for lr in range(P.local_rows(h, b, r, n)):
                gathered[r, lr * rb : (lr + 1) * rb] = img[P.global_row_of(lr, b, r, n)]
        assert torch.equal(P.assemble(gathered, h, rb, b, n), img)
