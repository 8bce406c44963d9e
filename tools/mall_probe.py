"""Is a weight matrix that sits in the 256 MiB Infinity Cache streamed faster than from HBM?  Times the skinny GEMMs of a
7B layer (hipGraph replay, no host floor) with ONE weight copy replayed back to back (warm: the matrix stays in the
memory-side cache when it fits) against 8+ copies in rotation (cold).  Decides whether prefetching the next GEMM's weights
into the Infinity Cache during the latency-bound attention phase could pay."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from triforce_amd import ops  # noqa: E402

DEV = "cuda:0"


def timeit(fns, iters=48):
    n = len(fns)
    for f in fns:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters):
            fns[i % n]()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


res = {}
for (name, N, K) in [("o", 4096, 4096), ("down", 4096, 11008), ("qkv", 12288, 4096), ("lm_head", 32000, 4096)]:
    x = torch.randn(8, K, device=DEV, dtype=torch.float16)
    copies = max(2, int(700e6 // (N * K * 2)) + 1)
    pls = [ops.PackedLinear(torch.randn(N, K, device=DEV, dtype=torch.float16) * 0.02) for _ in range(copies)]
    cold = timeit([(lambda p=p: ops.linear(x, p)) for p in pls])
    warm = timeit([lambda: ops.linear(x, pls[0])])
    mb = N * K * 2 / 1e6
    res[name] = {"MB": round(mb, 1), "cold_us": round(cold, 2), "warm_us": round(warm, 2),
                 "cold_GBps": round(mb / cold * 1e3, 1), "warm_GBps": round(mb / warm * 1e3, 1)}
    del pls
print(json.dumps(res), flush=True)
