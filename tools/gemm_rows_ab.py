"""Skinny decode GEMMs at 8 rows (one MFMA row tile) vs 17 rows (two: the gamma = 16 verifies), 7B and 13B widths; cold-cache
hipGraph chains as in tools/tune.py.  python tools/gemm_rows_ab.py > profiles/<file>.jsonl"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from triforce_amd import ops  # noqa: E402

DEV = "cuda:0"


def timeit(fns, iters=40):
    for f in fns:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters):
            fns[i % len(fns)]()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters * 1e3)
    return best


for model, hid, inter in (("7B", 4096, 11008), ("13B", 5120, 13824)):
    for name, N, K, split in (("qkv", 3 * hid, hid, 1), ("o", hid, hid, 1), ("gate_up", 2 * inter, hid, 2), ("down", hid, inter, 1)):
        copies = max(2, int(700e6 // (N * K * 2)) + 1)
        pls = [ops.PackedLinear(torch.randn(N, K, device=DEV, dtype=torch.float16) * 0.02, split=split) for _ in range(copies)]
        row = {"model": model, "gemm": name, "N": N, "K": K, "MB": round(N * K * 2 / 1e6, 1)}
        for M in ([1, 4, 8, 16] if os.environ.get("GEMM_ROWS_SMALL") else [8, 17, 32]):
            x = torch.randn(M, K, device=DEV, dtype=torch.float16)
            if split == 2:
                us = timeit([(lambda p=p: ops.mlp_act(x, p)) for p in pls])
            else:
                us = timeit([(lambda p=p: ops.linear(x, p)) for p in pls])
            row[f"M{M}_us"] = round(us, 2)
            row[f"M{M}_TBps"] = round(N * K * 2 / us / 1e6, 2)
        print(json.dumps(row), flush=True)
        del pls
