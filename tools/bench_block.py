"""Time the 128-row block attention (a prefill chunk) at a few cache lengths.  TRIFORCE_HIP_LIB selects a tuning build."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from triforce_amd import ops  # noqa: E402

DEV = "cuda:0"
H, D, sq = 32, 128, 128
g = torch.Generator(device=DEV).manual_seed(0)
res = {}
for sk in (16384, 62464, 124928):
    k = torch.randn(H, sk, D, generator=g, device=DEV, dtype=torch.float16)
    v = torch.randn(H, sk, D, generator=g, device=DEV, dtype=torch.float16)
    q = torch.randn(sq, H, D, generator=g, device=DEV, dtype=torch.float16)
    for _ in range(2):
        ops.attn_block(q, k, v, sk, 0.08837890625)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.attn_block(q, k, v, sk, 0.08837890625)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / 10 * 1e3
    res[sk] = {"us": round(us, 1), "TFLOPs": round(4 * sq * sk * H * D / us / 1e6, 1), "GBps": round(2 * sk * H * D * 2 / us / 1e3, 1)}
    del k, v
print(json.dumps(res))
