"""Split count of the split-KV attention at the head counts of a TP = 8 rank (4 / 5 heads): the default rule (one workgroup
per CU: 32..51 splits -> two-launch merge) against <= 8 splits with the merge inside the launch.  Cold-cache hipGraph chains
as in tools/attn_merge_ab.py.  python tools/nsplit_small_heads.py > profiles/<file>.jsonl"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from triforce_amd import ops  # noqa: E402

DEV = torch.device("cuda", 0)


def time_chain(H, sk, sq, nsplit, layers=32, reps=10):
    D = 128
    per = 2 * sk * H * D * 2
    nrep = max(2, min(layers, (1 << 30) // per + 1))
    g = torch.Generator(device=DEV).manual_seed(1)
    k = [torch.randn(H, sk, D, generator=g, device=DEV, dtype=torch.float16) for _ in range(nrep)]
    v = [torch.randn(H, sk, D, generator=g, device=DEV, dtype=torch.float16) for _ in range(nrep)]
    q = torch.randn(sq, H, D, generator=g, device=DEV, dtype=torch.float16)

    def chain():
        out = None
        for i in range(layers):
            out = ops.attn_decode(q, k[i % nrep], v[i % nrep], sk, 0.08837890625, nsplit=nsplit)
        return out
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        chain()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        chain()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / layers)
    ts.sort()
    return ts[len(ts) // 2], per


for label, H, sk, sq in [("7B TP8 rank retrieval verify", 4, 4103, 7), ("7B TP8 rank target verify", 4, 124935, 8),
                         ("13B TP8 rank retrieval verify g16", 5, 12305, 17), ("13B TP8 rank target verify g16", 5, 130066, 18),
                         ("7B TP4 rank retrieval verify", 8, 4103, 7), ("7B TP2 rank retrieval verify", 16, 4103, 7)]:
    row = {"shape": label, "H": H, "sk": sk, "sq": sq, "default_nsplit": ops._pick_nsplit(H, sk)}
    for ns in (None, 4, 8, 16, 32):
        us, per = time_chain(H, sk, sq, ns)
        row[f"nsplit_{ns or 'default'}_us"] = round(us, 2)
    print(json.dumps(row), flush=True)
