#!/usr/bin/env python3
"""Exercise, on ONE GPU with world_size = 1, the torch.distributed (RCCL) call pattern that
partition.DistributedRenderer uses for its gather — P2POp isend/irecv batches issued inside a side
stream that waits on an event of the compute stream, Work.wait() on the current stream — by sending
to self.  Not a substitute for a real N > 1 run, but it proves the API usage and stream plumbing."""
import os
import sys

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fractal_renderer_amd as fr  # noqa: E402
from fractal_renderer_amd import partition as P  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
fr.init(0)
dev = torch.device("cuda", 0)
cfg = fr.Config.new()
cfg.width, cfg.height, cfg.iterations = 1024, 768, 200
cfg.pos.re, cfg.exposure = -0.6, 5.0
rb = 3 * cfg.width
compute = torch.cuda.current_stream(dev)
comm = torch.cuda.Stream(dev)
src = torch.zeros(cfg.height * rb, dtype=torch.uint8, device=dev)
dst = torch.zeros_like(src)
works = []
B = 256
for b in range(P.num_blocks(cfg.height, B)):
    y0, y1 = P.block_range(cfg.height, B, b)
    P.render_rows_hip(cfg, 0, y0, y1, src[y0 * rb : y1 * rb], compute.cuda_stream)
    ev = torch.cuda.Event()
    ev.record(compute)
    comm.wait_event(ev)
    with torch.cuda.stream(comm):
        ops = [dist.P2POp(dist.irecv, dst[y0 * rb : y1 * rb], 0), dist.P2POp(dist.isend, src[y0 * rb : y1 * rb], 0)]
        works.extend(dist.batch_isend_irecv(ops))
for w in works:
    w.wait()
compute.wait_stream(comm)
torch.cuda.synchronize()
ref = torch.from_numpy(fr.get_image(cfg)).to(dev).view(-1)
ok = bool(torch.equal(dst, ref)) and bool(torch.equal(src, ref))
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.destroy_process_group()
print("nccl p2p self-test:", "ok" if ok else "MISMATCH", "| works:", len(works), "| all_reduce:", float(t.item()))
sys.exit(0 if ok else 1)
