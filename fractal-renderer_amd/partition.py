"""Row-block-cyclic partition of one image over the GPUs of a node, and its reassembly.

The reference parallelises get_image over ROWS (src/lib.rs:256-258) and concatenates them in
index order (:266-267); pixels are independent, so rows shard with no exchange except the final
gather of the finished bytes on the root.  Contiguous bands would be badly unbalanced (the set's
interior sits in the middle rows of the default view), so rank r renders row blocks
r, r + world, r + 2*world, ... of `block_rows` rows each and packs them contiguously.

torch is used here only as plumbing (device buffers, streams, torch.distributed = RCCL on ROCm,
gloo in the CPU tests); the rendering itself goes through the C ABI.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _native

DEFAULT_BLOCK_ROWS = 64


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


def render_local_hip(config, precision, block_rows, rank, world, out, stream_ptr):
    """Render this rank's share into `out` (uint8 CUDA tensor, >= 3*width*local_rows bytes)
    asynchronously on the HIP stream `stream_ptr`.  Returns the rows written."""
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
    uint8 [height, row_bytes] in image row order.  Pure tensor indexing; device-agnostic."""
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


def gather_to_root(local, height, row_bytes, block_rows, rank, world, group=None, scratch=None):
    """Final gather of the rendered rows (RCCL over xGMI on a GPU node): every rank sends its
    packed rows to rank 0, which returns the assembled image [height, row_bytes]; other ranks
    return None.  `local` holds at least local_rows * row_bytes bytes."""
    if world == 1:
        return local[: height * row_bytes].view(height, row_bytes)
    max_rows = local_rows(height, block_rows, 0, world)  # rank 0 always owns the most rows
    n = max_rows * row_bytes
    send = local[:n]
    if send.numel() < n:  # a rank with fewer rows: pad the send buffer to the common size
        padded = torch.empty(n, dtype=torch.uint8, device=local.device)
        padded[: local.numel()] = local
        send = padded
    if rank == 0:
        if scratch is None or scratch.numel() < world * n:
            scratch = torch.empty(world * n, dtype=torch.uint8, device=local.device)
        gathered = scratch[: world * n].view(world, n)
        dist.gather(send, gather_list=[gathered[r] for r in range(world)], dst=0, group=group)
        return assemble(gathered, height, row_bytes, block_rows, world)
    dist.gather(send, gather_list=None, dst=0, group=group)
    return None


def render_distributed(config, precision=0, block_rows=DEFAULT_BLOCK_ROWS, rank=None, world=None, group=None,
                       render_fn=None, device=None):
    """get_image (src/lib.rs:253-270) across `world` ranks: returns uint8 [height, width, 3] on
    rank 0, None elsewhere.  `render_fn(config, precision, block_rows, rank, world, out)` may
    replace the HIP renderer (the gloo tests inject a CPU stand-in to exercise the partition)."""
    rank = dist.get_rank(group) if rank is None else rank
    world = dist.get_world_size(group) if world is None else world
    row_bytes = 3 * config.width
    max_rows = local_rows(config.height, block_rows, 0, world)
    if render_fn is None:
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        local = torch.empty(max(max_rows * row_bytes, 1), dtype=torch.uint8, device=device)
        stream = torch.cuda.current_stream(device).cuda_stream
        render_local_hip(config, precision, block_rows, rank, world, local, stream)
    else:
        local = torch.empty(max(max_rows * row_bytes, 1), dtype=torch.uint8, device=device or "cpu")
        render_fn(config, precision, block_rows, rank, world, local)
    img = gather_to_root(local, config.height, row_bytes, block_rows, rank, world, group)
    if img is None:
        return None
    return img.view(config.height, config.width, 3)
