"""Row-block-cyclic partition of one image over the GPUs of a node, and the final gather.

The reference parallelises get_image over ROWS (src/lib.rs:256-258) and concatenates them in
index order (:266-267); pixels are independent, so rows shard with no exchange except the final
gather of the finished bytes on the root.  Contiguous bands would be badly unbalanced (the set's
interior sits in the middle rows of the default view), so rank r renders row blocks
r, r + world, r + 2*world, ... of `block_rows` rows each.

Gather = point-to-point sends of finished blocks to rank 0 (RCCL over xGMI on a GPU node: each
peer -> root transfer rides its own link, so the 7 peers of an 8-GPU node land in parallel).  Round
k of the schedule is "block k*world + r from every rank r": those blocks are ADJACENT in the image,
so rank 0 receives every block straight into its final place — no reorder pass — and renders its
own blocks in place.  A rank renders several of its blocks per kernel launch ("chunk": big chunks
first, a geometric tail 4, 2, 1 last) and chunks are pipelined: chunk c is on the wire (communication
stream) while chunk c+1 renders (two alternating compute streams).

torch is used here only as plumbing (device buffers, streams, torch.distributed = RCCL on ROCm,
gloo in the CPU tests); the rendering itself goes through the C ABI.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _native

DEFAULT_BLOCK_ROWS = 256
N_BLOCK_STREAMS = 2      # chunk kernels alternate between this many streams (measured: 1 stream costs +60 %
                         # at 256-row launches because nothing covers a kernel's tail; 4 is no better than 2)
MAX_CHUNK_BLOCKS = 8     # blocks per launch (2048 rows at the default block size: as fast as one launch)


def chunk_schedule(nb):
    """Local block index ranges [j0, j1) a rank renders per launch: big chunks first (launches of
    >= 1024 rows run at the single-launch rate; 256-row launches measured 17 % slower), then a
    geometric tail 4, 2, 1 so that the bytes still to be sent when the last kernel ends are one block.
    A chunk is at most a quarter of the rank's blocks: a rank's transfer takes about as long as its
    rendering (both scale as 1/N), so the link has to start early and stay busy — with 8 blocks per rank
    (16384^2 over 8 GPUs) that is 1, 2, 2, 2, 1 rather than 1, 4, 2, 1."""
    cap = max(1, min(MAX_CHUNK_BLOCKS, nb // 4))
    sizes, rem, s = [], nb, 1
    while rem > 0:
        t = min(s, rem, cap)
        sizes.append(t)
        rem -= t
        s *= 2
    sizes.reverse()
    ranges, j = [], 0
    for t in sizes:
        ranges.append((j, j + t))
        j += t
    return ranges


def num_blocks(height, block_rows):
    return (height + block_rows - 1) // block_rows


def shares(rank, world, root_share=1):
    """The row blocks of rank `rank` as arithmetic progressions (first_block, stride) — what one launch of the C ABI's
    block-cyclic entry points renders.  root_share = 1 (default): plain cyclic dealing, block b -> rank b % world.
    root_share = q in {2, 4}: the sink (rank 0) keeps only every q-th of ITS blocks — a 1/q share — and the rest of them are
    dealt round-robin to the other ranks, each of which then holds its own progression plus q - 1 short ones;
    root_share = 0: the sink renders nothing and only receives.  (Whether the sink should render less is a question for the
    first real multi-GPU run — DESIGN.md §4; the bytes are the same whatever the dealing.)"""
    q = int(root_share)
    if q not in (0, 1, 2, 4):
        raise ValueError("root_share must be 1 (full share), 2, 4 (a half, a quarter) or 0 (the sink only receives)")
    if world == 1 or q == 1:
        return [(rank, world)]
    peers = world - 1
    if rank == 0:
        return [] if q == 0 else [(0, q * world)]
    out = [(rank, world)]
    if q == 0:  # every block of the sink: block j * world, j = 0, 1, ... -> peer 1 + j % peers
        out.append(((rank - 1) * world, peers * world))
    else:       # the sink's blocks j * world with j % q == c, c = 1 .. q - 1: the i-th of them -> peer 1 + i % peers
        for c in range(1, q):
            out.append(((c + q * (rank - 1)) * world, q * peers * world))
    return out


def share_blocks(rank, world, nblocks, root_share=1):
    """Per progression of shares(): the global block indices below nblocks, in rendering order."""
    return [list(range(first, nblocks, stride)) for first, stride in shares(rank, world, root_share)]


def rank_chunks(rank, world, nblocks, root_share=1):
    """The launches of one rank for one image: a list of chunks, each a list of global block indices forming an arithmetic
    progression (one call of render_chunk_hip).  Every rank can compute every rank's list: step s of the gather moves
    chunk s of every peer."""
    out = []
    for (first, stride), blocks in zip(shares(rank, world, root_share), share_blocks(rank, world, nblocks, root_share)):
        # the schedule of the LONGEST progression of this stride (the one starting at block 0), so that ranks whose
        # progression is one block shorter cut it at the same places
        for j0, j1 in chunk_schedule((nblocks + stride - 1) // stride):
            if blocks[j0:j1]:
                out.append(blocks[j0:j1])
    return out


def block_range(height, block_rows, b):
    """Image rows [y0, y1) of block b."""
    y0 = b * block_rows
    return y0, min(height, y0 + block_rows)


def local_rows(height, block_rows, rank, world, root_share=1):
    """Rows rank `rank` renders (per progression the arithmetic of fr_block_cyclic_rows in the C ABI)."""
    rows = 0
    for blocks in share_blocks(rank, world, num_blocks(height, block_rows), root_share):
        for b in blocks:
            rows += min(block_rows, height - b * block_rows)
    return rows


def global_row_of(local_row, block_rows, rank, world):
    """Image row of a rank's packed local row."""
    return ((local_row // block_rows) * world + rank) * block_rows + local_row % block_rows


def render_rows_hip(config, precision, y0, y1, out, stream_ptr):
    """Rows [y0, y1) into `out` (uint8 CUDA tensor view, exactly 3*width*(y1-y0) bytes), async on
    the HIP stream `stream_ptr`."""
    _native.check(
        _native.load().fr_render_rows_rgb8_device(
            C.byref(config), int(precision), y0, y1, out.data_ptr(), out.numel(), stream_ptr
        )
    )


def render_chunk_hip(config, precision, block_rows, first_block, block_stride, max_blocks, in_place, out, stream_ptr):
    """Blocks first_block, first_block + stride, ... (at most max_blocks) in ONE launch: packed into
    `out`, or — in_place — each row at its place in the whole image whose base is `out`."""
    rows = C.c_uint64(0)
    _native.check(
        _native.load().fr_render_block_cyclic_range_rgb8_device(
            C.byref(config), int(precision), block_rows, first_block, block_stride, max_blocks, 1 if in_place else 0,
            out.data_ptr(), out.numel(), stream_ptr, C.byref(rows),
        )
    )
    return rows.value


def render_local_hip(config, precision, block_rows, rank, world, out, stream_ptr, root_share=1):
    """Render this rank's whole share, packed, in ONE launch per progression (used when nothing is gathered)."""
    total, off = 0, 0
    for first, stride in shares(rank, world, root_share):
        rows = C.c_uint64(0)
        part = out[off:]
        _native.check(
            _native.load().fr_render_block_cyclic_rgb8_device(
                C.byref(config), int(precision), block_rows, first, stride, part.data_ptr(), part.numel(), stream_ptr,
                C.byref(rows),
            )
        )
        total += rows.value
        off += rows.value * 3 * config.width
    return total


def assemble(gathered, height, row_bytes, block_rows, world):
    """gathered: uint8 [world, max_local_rows * row_bytes] (rank r's packed rows first) ->
    uint8 [height, row_bytes] in image row order.  Pure tensor indexing; device-agnostic.
    (Reference implementation of the layout; the pipelined gather below never needs it.)"""
    img = torch.empty((height, row_bytes), dtype=torch.uint8, device=gathered.device)
    full_blocks = height // block_rows
    tail = height - full_blocks * block_rows
    img_blocks = img[: full_blocks * block_rows].view(full_blocks, block_rows * row_bytes)
    for r in range(world):
        nb = len(range(r, full_blocks, world))
        if nb:
            img_blocks[r::world] = gathered[r, : nb * block_rows * row_bytes].view(nb, block_rows * row_bytes)
        if tail and full_blocks % world == r:
            src = gathered[r, nb * block_rows * row_bytes : (nb * block_rows + tail) * row_bytes]
            img[full_blocks * block_rows :] = src.view(tail, row_bytes)
    return img


class DistributedRenderer:
    """get_image (src/lib.rs:253-270) across the ranks of a process group, result on rank 0.

    Buffers are allocated once and reused by every render() call (HBM-resident): rank 0 holds
    the full image [height, 3*width]; other ranks hold their blocks packed.  `render_rows` may
    replace the HIP renderer (the gloo CPU tests inject a stand-in): it is called as
    render_rows(config, precision, y0, y1, out_view)."""

    def __init__(self, config, precision=0, block_rows=DEFAULT_BLOCK_ROWS, group=None, device=None,
                 render_rows=None, force_blocks=False, root_share=1):
        self.config = config
        self.root_share = int(root_share)
        self.timing = None  # set to a dict by render(diagnose=True): per-chunk device events of ONE step
        self.precision = int(precision)
        self.block_rows = int(block_rows)
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.row_bytes = 3 * config.width
        self.height = config.height
        self.nblocks = num_blocks(self.height, self.block_rows)
        self.cuda = render_rows is None
        if self.cuda and self.block_rows % 8 != 0:
            raise ValueError("block_rows must be a multiple of 8 (rank 0 renders its blocks in place, tile by tile)")
        self.force_blocks = bool(force_blocks)  # world == 1: still render block by block (tests, tuning)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if self.cuda else torch.device("cpu")
        self.device = device
        self._render_rows = render_rows
        my = local_rows(self.height, self.block_rows, self.rank, self.world, self.root_share)
        if self.rank == 0:
            self.image = torch.empty(max(self.height * self.row_bytes, 1), dtype=torch.uint8, device=device)
            self.local = None
        else:
            self.image = None
            self.local = torch.empty(max(my * self.row_bytes, 1), dtype=torch.uint8, device=device)
        if self.cuda:
            self.compute_stream = torch.cuda.current_stream(device)
            # consecutive blocks render on alternating streams so that the tail of one block's kernel
            # (its last, longest strips) overlaps the start of the next instead of idling the GPU
            self.block_streams = [self.compute_stream] + [torch.cuda.Stream(device) for _ in range(N_BLOCK_STREAMS - 1)]
            self.comm_stream = torch.cuda.Stream(device) if self.world > 1 else None
        else:
            self.compute_stream = self.comm_stream = None
            self.block_streams = [None] * N_BLOCK_STREAMS

    # -- helpers ---------------------------------------------------------------------------
    def _image_rows(self, y0, y1):
        return self.image[y0 * self.row_bytes : y1 * self.row_bytes]

    def _render(self, y0, y1, out, stream=None):
        if self.cuda:
            stream = self.compute_stream if stream is None else stream
            render_rows_hip(self.config, self.precision, y0, y1, out, stream.cuda_stream)
        else:
            self._render_rows(self.config, self.precision, y0, y1, out)

    # -- the step -------------------------------------------------------------------------
    def render(self, diagnose=False):
        """One full image.  Returns the [height, width, 3] uint8 image on rank 0, None elsewhere.
        On CUDA the call is asynchronous w.r.t. the host except for torch.distributed's own
        bookkeeping; call torch.cuda.synchronize() (or use the result on the current stream,
        which is made to wait for the gather) before reading.
        diagnose=True (CUDA): device events around every chunk's kernel and every step's transfers are kept in
        self.timing for step_report() — what bench.py prints per rank for an N > 1 run."""
        cfg, B, N, r = self.config, self.block_rows, self.world, self.rank
        if N == 1 and not self.force_blocks:
            # single launch of the whole image, rendered in place
            self._render(0, self.height, self._image_rows(0, self.height))
            return self.image[: self.height * self.row_bytes].view(self.height, cfg.width, 3)

        works = []
        local_off = 0
        timing = {"kernels": [], "transfers": [], "bytes_sent": 0, "bytes_received": 0, "blocks": 0} if (diagnose and self.cuda) else None
        if self.cuda:
            for st in self.block_streams[1:]:
                st.wait_stream(self.compute_stream)  # order behind earlier work
            if timing is not None:
                timing["t0"] = torch.cuda.Event(enable_timing=True)
                timing["t0"].record(self.compute_stream)
        # every rank's launches, step by step: step s renders chunk s of this rank and moves chunk s of every peer
        chunks = [rank_chunks(q, N, self.nblocks, self.root_share) for q in range(N)]
        nsteps = max(len(c) for c in chunks)
        for s in range(nsteps):
            mine = chunks[r][s] if s < len(chunks[r]) else []
            done_event, sends = None, []
            if mine:
                stream = self.block_streams[s % N_BLOCK_STREAMS]
                stride = mine[1] - mine[0] if len(mine) > 1 else 1
                if timing is not None:
                    k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    k0.record(stream)
                    timing["kernels"].append((k0, k1))
                    timing["blocks"] += len(mine)
                if r == 0:
                    # rank 0 renders every chunk in place
                    if self.cuda:
                        render_chunk_hip(cfg, self.precision, B, mine[0], stride, len(mine), True, self.image,
                                         stream.cuda_stream)
                    else:
                        for b in mine:
                            y0, y1 = block_range(self.height, B, b)
                            self._render(y0, y1, self._image_rows(y0, y1))
                else:
                    rows = sum(block_range(self.height, B, b)[1] - block_range(self.height, B, b)[0] for b in mine)
                    chunk = self.local[local_off : local_off + rows * self.row_bytes]
                    if self.cuda:
                        render_chunk_hip(cfg, self.precision, B, mine[0], stride, len(mine), False, chunk, stream.cuda_stream)
                    off = 0
                    for b in mine:
                        y0, y1 = block_range(self.height, B, b)
                        part = chunk[off : off + (y1 - y0) * self.row_bytes]
                        if not self.cuda:
                            self._render(y0, y1, part)
                        sends.append(part)
                        off += (y1 - y0) * self.row_bytes
                    local_off += rows * self.row_bytes
                if self.cuda:
                    if timing is not None:
                        timing["kernels"][-1][1].record(stream)
                    done_event = torch.cuda.Event()
                    done_event.record(stream)
            if N == 1:
                continue
            # hand step s to the communication stream; the next chunk renders meanwhile
            if self.cuda:
                if done_event is not None and r != 0:
                    self.comm_stream.wait_event(done_event)  # a send needs its chunk; rank 0's receives do not
                ctx = torch.cuda.stream(self.comm_stream)
            else:
                ctx = _NullContext()
            with ctx:
                if r == 0:
                    ops = []
                    for src in range(1, N):  # per peer, in the peer's own sending order
                        for b in (chunks[src][s] if s < len(chunks[src]) else []):
                            y0, y1 = block_range(self.height, B, b)
                            ops.append(dist.P2POp(dist.irecv, self._image_rows(y0, y1), src, self.group))
                            if timing is not None:
                                timing["bytes_received"] += (y1 - y0) * self.row_bytes
                else:
                    ops = [dist.P2POp(dist.isend, part, 0, self.group) for part in sends]
                    if timing is not None:
                        timing["bytes_sent"] += sum(p.numel() for p in sends)
                if ops:
                    if timing is not None:
                        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        c0.record(self.comm_stream)
                    works.extend(dist.batch_isend_irecv(ops))
                    if timing is not None:
                        c1.record(self.comm_stream)
                        timing["transfers"].append((c0, c1))
        for w in works:
            w.wait()  # CUDA: makes the current stream wait for the transfer; gloo: blocks
        if self.cuda:
            for st in self.block_streams[1:]:
                self.compute_stream.wait_stream(st)
            if self.comm_stream is not None:
                self.compute_stream.wait_stream(self.comm_stream)
            if timing is not None:
                timing["t1"] = torch.cuda.Event(enable_timing=True)
                timing["t1"].record(self.compute_stream)
                self.timing = timing
        if r == 0:
            return self.image[: self.height * self.row_bytes].view(self.height, cfg.width, 3)
        return None

    def step_report(self):
        """After render(diagnose=True) and a device synchronisation: this rank's step in milliseconds of DEVICE time —
        kernel_ms (its chunk kernels, summed), compute_span_ms (first kernel start to last kernel end), transfer_ms (its
        grouped P2P calls on the communication stream, summed: sends on a peer, receives on the sink), transfer_span_ms,
        step_ms (start of the step to everything joined on the compute stream), idle_ms = step - compute span, and the
        bytes it sent / received.  bench.py gathers these from every rank."""
        t = self.timing
        if not t:
            return None

        def span(pairs):
            if not pairs:
                return 0.0, 0.0
            total = sum(a.elapsed_time(b) for a, b in pairs)
            return total, max(t["t0"].elapsed_time(b) for _, b in pairs) - min(t["t0"].elapsed_time(a) for a, _ in pairs)

        k_sum, k_span = span(t["kernels"])
        c_sum, c_span = span(t["transfers"])
        step = t["t0"].elapsed_time(t["t1"])
        return {"rank": self.rank, "blocks": t["blocks"], "launches": len(t["kernels"]), "kernel_ms": k_sum, "compute_span_ms": k_span,
                "transfer_ms": c_sum, "transfer_span_ms": c_span, "step_ms": step, "idle_ms": max(step - k_span, 0.0),
                "bytes_sent": t["bytes_sent"], "bytes_received": t["bytes_received"]}


class _NullContext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def render_distributed(config, precision=0, block_rows=DEFAULT_BLOCK_ROWS, group=None, render_rows=None,
                       device=None, root_share=1):
    """One-shot convenience wrapper around DistributedRenderer."""
    return DistributedRenderer(config, precision, block_rows, group, device, render_rows, root_share=root_share).render()
