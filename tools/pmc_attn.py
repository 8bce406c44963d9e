"""Run ONLY the target-verify attention at BASELINE configs[1] shape a few times, for rocprofv3 --pmc passes
(FETCH_SIZE / WRITE_SIZE in separate runs; MI355X_MICROARCH.md §HBM: double FETCH_SIZE on gfx950)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from triforce_amd import ops  # noqa: E402

DEV = "cuda:0"
sq, sk, H, D = 8, 124936, 32, 128
g = torch.Generator(device=DEV).manual_seed(0)
k = torch.randn(H, sk, D, generator=g, device=DEV, dtype=torch.float16)
v = torch.randn(H, sk, D, generator=g, device=DEV, dtype=torch.float16)
q = torch.randn(sq, H, D, generator=g, device=DEV, dtype=torch.float16)
# a second, different KV pair so consecutive launches cannot be served from the 256 MiB Infinity Cache
k2, v2 = k.flip(1).contiguous(), v.flip(1).contiguous()
for i in range(6):
    ops.attn_decode(q, k if i % 2 == 0 else k2, v if i % 2 == 0 else v2, sk, 0.08837890625)
torch.cuda.synchronize()
print("algorithmic bytes per launch:", 2 * sk * H * D * 2)
