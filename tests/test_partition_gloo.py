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


def _worker(rank, world, port, w, h, block_rows, q, root_share=1):
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
        renderer = P.DistributedRenderer(cfg, 0, block_rows, render_rows=_oracle_render_rows(ocfg), root_share=root_share)
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


@pytest.mark.parametrize("world,w,h,block_rows,root_share", [(2, 16, 70, 8, 2), (3, 20, 101, 4, 4), (4, 12, 203, 8, 0), (8, 10, 1003, 8, 2),
                                                              (3, 9, 5, 8, 0)])
def test_the_sink_may_render_a_smaller_share(world, w, h, block_rows, root_share):
    """rank 0 is both a renderer and the sink of the gather; root_share deals part (1/2, 1/4) or all (0) of
    ITS blocks to the other ranks — each then holds several arithmetic progressions of blocks, every chunk still one launch
    of the C ABI — and the image must come out the same."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, w, h, block_rows, q, root_share)) for r in range(world)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) is True


def test_every_dealing_covers_every_block_exactly_once_in_progressions():
    from fractal_renderer_amd import partition as P

    for world in (1, 2, 3, 4, 8, 16):
        for q in (1, 2, 4, 0):
            for nb in (1, 7, 13, 64, 256):
                owner = {}
                for r in range(world):
                    for ch in P.rank_chunks(r, world, nb, q):
                        assert ch and all(ch[i + 1] - ch[i] == ch[1] - ch[0] for i in range(len(ch) - 1))  # one launch each
                        for b in ch:
                            assert b not in owner and 0 <= b < nb
                            owner[b] = r
                assert sorted(owner) == list(range(nb)), (world, q, nb)
                if world > 1 and q == 0:
                    assert 0 not in owner.values()
                if world > 1 and q in (2, 4) and nb >= 4 * world * q:
                    mine = sum(1 for v in owner.values() if v == 0)
                    assert abs(mine - nb / (world * q)) <= 1, (world, q, nb, mine)
    # plain cyclic dealing is what it always was: block b -> rank b % world, the same cuts for every rank
    assert P.rank_chunks(1, 8, 64, 1) == [[1], [9, 17], [25, 33], [41, 49], [57]]
    assert P.rank_chunks(7, 8, 63, 1) == [[7], [15, 23], [31, 39], [47, 55]]


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
