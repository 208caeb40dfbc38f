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
own blocks in place.  Rounds are pipelined: round k is on the wire (communication stream) while
round k+1 renders (compute stream).

torch is used here only as plumbing (device buffers, streams, torch.distributed = RCCL on ROCm,
gloo in the CPU tests); the rendering itself goes through the C ABI.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _native

DEFAULT_BLOCK_ROWS = 256


def num_blocks(height, block_rows):
    return (height + block_rows - 1) // block_rows


def block_range(height, block_rows, b):
    """Image rows [y0, y1) of block b."""
    y0 = b * block_rows
    return y0, min(height, y0 + block_rows)


def local_rows(height, block_rows, rank, world):
    """Rows rank `rank` renders (same arithmetic as fr_block_cyclic_rows in the C ABI)."""
    rows = 0
    b = rank
    while b * block_rows < height:
        rows += min(block_rows, height - b * block_rows)
        b += world
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


def render_local_hip(config, precision, block_rows, rank, world, out, stream_ptr):
    """Render this rank's whole share, packed, in ONE launch (used when nothing is gathered)."""
    rows = C.c_uint64(0)
    _native.check(
        _native.load().fr_render_block_cyclic_rgb8_device(
            C.byref(config), int(precision), block_rows, rank, world, out.data_ptr(), out.numel(), stream_ptr,
            C.byref(rows),
        )
    )
    return rows.value


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
                 render_rows=None):
        self.config = config
        self.precision = int(precision)
        self.block_rows = int(block_rows)
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.row_bytes = 3 * config.width
        self.height = config.height
        self.nblocks = num_blocks(self.height, self.block_rows)
        self.cuda = render_rows is None
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if self.cuda else torch.device("cpu")
        self.device = device
        self._render_rows = render_rows
        my = local_rows(self.height, self.block_rows, self.rank, self.world)
        if self.rank == 0:
            self.image = torch.empty(max(self.height * self.row_bytes, 1), dtype=torch.uint8, device=device)
            self.local = None
        else:
            self.image = None
            self.local = torch.empty(max(my * self.row_bytes, 1), dtype=torch.uint8, device=device)
        if self.cuda:
            self.compute_stream = torch.cuda.current_stream(device)
            self.comm_stream = torch.cuda.Stream(device) if self.world > 1 else None
        else:
            self.compute_stream = self.comm_stream = None

    # -- helpers ---------------------------------------------------------------------------
    def _image_rows(self, y0, y1):
        return self.image[y0 * self.row_bytes : y1 * self.row_bytes]

    def _render(self, y0, y1, out):
        if self.cuda:
            render_rows_hip(self.config, self.precision, y0, y1, out, self.compute_stream.cuda_stream)
        else:
            self._render_rows(self.config, self.precision, y0, y1, out)

    # -- the step -------------------------------------------------------------------------
    def render(self):
        """One full image.  Returns the [height, width, 3] uint8 image on rank 0, None elsewhere.
        On CUDA the call is asynchronous w.r.t. the host except for torch.distributed's own
        bookkeeping; call torch.cuda.synchronize() (or use the result on the current stream,
        which is made to wait for the gather) before reading."""
        cfg, B, N, r = self.config, self.block_rows, self.world, self.rank
        if N == 1:
            # single launch of the whole image, rendered in place
            self._render(0, self.height, self._image_rows(0, self.height))
            return self.image[: self.height * self.row_bytes].view(self.height, cfg.width, 3)

        works = []
        rounds = (self.nblocks + N - 1) // N
        local_off = 0
        for k in range(rounds):
            b_mine = k * N + r
            if b_mine < self.nblocks:
                y0, y1 = block_range(self.height, B, b_mine)
                nbytes = (y1 - y0) * self.row_bytes
                if r == 0:
                    out = self._image_rows(y0, y1)  # rank 0 renders in place
                else:
                    out = self.local[local_off : local_off + nbytes]
                    local_off += nbytes
                self._render(y0, y1, out)
            else:
                out = None
            # hand round k to the communication stream; round k+1 renders meanwhile
            if self.cuda:
                self.comm_stream.wait_stream(self.compute_stream)
                ctx = torch.cuda.stream(self.comm_stream)
            else:
                ctx = _NullContext()
            with ctx:
                if r == 0:
                    ops = []
                    for src in range(1, N):
                        b = k * N + src
                        if b < self.nblocks:
                            y0, y1 = block_range(self.height, B, b)
                            ops.append(dist.P2POp(dist.irecv, self._image_rows(y0, y1), src, self.group))
                    if ops:
                        works.extend(dist.batch_isend_irecv(ops))
                elif out is not None:
                    works.extend(dist.batch_isend_irecv([dist.P2POp(dist.isend, out, 0, self.group)]))
        for w in works:
            w.wait()  # CUDA: makes the current stream wait for the transfer; gloo: blocks
        if self.cuda:
            self.compute_stream.wait_stream(self.comm_stream)
        if r == 0:
            return self.image[: self.height * self.row_bytes].view(self.height, cfg.width, 3)
        return None


class _NullContext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def render_distributed(config, precision=0, block_rows=DEFAULT_BLOCK_ROWS, group=None, render_rows=None,
                       device=None):
    """One-shot convenience wrapper around DistributedRenderer."""
    return DistributedRenderer(config, precision, block_rows, group, device, render_rows).render()
