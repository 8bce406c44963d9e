"""Run the kernels of ONE retrieval-verify decoder layer at BASELINE configs[1] shapes (7B widths, 7 rows, 4 103
retrieval slots) a few times each, for rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE in separate runs) — the HBM
traffic of the stage that is 47 % of the decode step, kernel by kernel, next to its algorithmic bytes.  Every launch
uses the next of several weight / KV copies (> 600 MB per shape) so the 256 MiB Infinity Cache cannot serve re-reads.
The launch list (label + algorithmic bytes, in launch order) goes to gpurun_out/pmc_layer_meta.json for
tools/pmc_layer_reduce.py.
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir> -- python tools/pmc_layer.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from triforce_amd import ops  # noqa: E402

DEV = "cuda:0"
ROWS, HID, INTER, H, D, SLOTS = 7, 4096, 11008, 32, 128, 4103
REPS = 6
meta = []
g = torch.Generator(device=DEV).manual_seed(0)


def copies(n, k, split=1):
    c = max(2, int(700e6 // (n * k * 2)) + 1)
    return [ops.PackedLinear(torch.randn(n, k, generator=g, device=DEV, dtype=torch.float16) * 0.02, split=split)
            for _ in range(c)]


x = torch.randn(ROWS, HID, generator=g, device=DEV, dtype=torch.float16)
act = torch.randn(ROWS, INTER, generator=g, device=DEV, dtype=torch.float16)
for label, n, k, inp in (("q|k|v GEMM", 3 * HID, HID, x), ("o_proj GEMM", HID, HID, x), ("down_proj GEMM", HID, INTER, act)):
    pls = copies(n, k)
    torch.cuda.synchronize()
    for i in range(REPS):
        ops.linear(inp, pls[i % len(pls)])
        meta.append({"label": label, "kernel": "skinny_gemm", "algorithmic_bytes": n * k * 2})
    torch.cuda.synchronize()
    del pls
pls = copies(2 * INTER, HID, split=2)
torch.cuda.synchronize()
for i in range(REPS):
    ops.mlp_act(x, pls[i % len(pls)])
    meta.append({"label": "gate|up GEMM + SwiGLU", "kernel": "skinny_gemm", "algorithmic_bytes": 2 * INTER * HID * 2})
torch.cuda.synchronize()
del pls
kvs = [(torch.randn(H, SLOTS, D, generator=g, device=DEV, dtype=torch.float16),
        torch.randn(H, SLOTS, D, generator=g, device=DEV, dtype=torch.float16)) for _ in range(12)]
q = torch.randn(ROWS, H, D, generator=g, device=DEV, dtype=torch.float16)
torch.cuda.synchronize()
for i in range(REPS):
    k_, v_ = kvs[i % len(kvs)]
    ops.attn_decode(q, k_, v_, SLOTS, 0.08837890625)
    meta.append({"label": "retrieval-verify attention", "kernel": "attn_split", "algorithmic_bytes": 2 * SLOTS * H * D * 2})
torch.cuda.synchronize()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "pmc_layer_meta.json"), "w") as f:
    json.dump({"rows": ROWS, "hidden": HID, "inter": INTER, "heads": H, "head_dim": D, "slots": SLOTS, "launches": meta}, f)
print("launches:", len(meta))
