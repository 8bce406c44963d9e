"""Probe: can an RCCL all-reduce issued through torch.distributed be captured in a hipGraph and replayed?
Run with 2+ ranks (torchrun).  With PROBE_SAME_GPU=1 every rank uses cuda:0 (two processes on one GPU) — only a
plumbing check for boxes with a single GPU, if RCCL accepts duplicate devices at all."""
import os
import sys

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = 0 if os.environ.get("PROBE_SAME_GPU") == "1" else int(os.environ.get("LOCAL_RANK", rank))
torch.cuda.set_device(dev)
dist.init_process_group("nccl")
x = torch.full((7, 4096), float(rank + 1), dtype=torch.float16, device="cuda")
dist.all_reduce(x)
torch.cuda.synchronize()
print(f"[rank {rank}] eager all_reduce ok: {float(x[0, 0])} (expect {world * (world + 1) / 2})", flush=True)
static = torch.full((7, 4096), float(rank + 1), dtype=torch.float16, device="cuda")
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        y = static * 2
        dist.all_reduce(y)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode=os.environ.get("PROBE_CAPTURE_MODE", "thread_local")):
    y = static * 2
    dist.all_reduce(y)
    z = y + 1
torch.cuda.synchronize()
for it in range(3):
    static.fill_(float(rank + 1 + it))
    g.replay()
    torch.cuda.synchronize()
    want = 2 * sum(r + 1 + it for r in range(world)) + 1
    print(f"[rank {rank}] replay {it}: {float(z[0, 0])} (expect {want})", flush=True)
    assert float(z[0, 0]) == want
dist.barrier()
dist.destroy_process_group()
print(f"[rank {rank}] RCCL capture probe PASSED", flush=True)
