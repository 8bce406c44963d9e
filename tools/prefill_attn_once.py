"""One tf_attn_prefill launch series for counter collection (rocprofv3 --pmc ... -- python tools/prefill_attn_once.py [sk])."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from triforce_amd import ops  # noqa: E402

DEV = torch.device("cuda", 0)
sk = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
H, D, chunk = 32, 128, 1024
g = torch.Generator(device=DEV).manual_seed(1)
k = torch.randn(H, sk, D, generator=g, device=DEV, dtype=torch.float16)
v = torch.randn(H, sk, D, generator=g, device=DEV, dtype=torch.float16)
q = torch.randn(chunk, H, D, generator=g, device=DEV, dtype=torch.float16)
for _ in range(3):
    ops.attn_prefill(q, k, v, sk, D ** -0.5)
torch.cuda.synchronize()
