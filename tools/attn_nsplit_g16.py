"""Round 4, verdict item 1b/1c: split count of the two-q-tile split-KV attention at the gamma = 16 shapes (17 / 18 rows) of
configs[3] / configs[4] — H = 40 (13B on one GPU), 32, 16 (7B TP 2 rank), 5 (13B TP 8 rank) — and the retrieval scorer at
those head counts.  Cold-cache hipGraph chains (tools/nsplit_small_heads.py).

    python tools/attn_nsplit_g16.py > profiles/r04_attn_nsplit_g16.jsonl
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from triforce_amd import ops  # noqa: E402

DEV = torch.device("cuda", 0)


def chain_time(fn_of_i, layers=32, reps=7):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(layers):
            fn_of_i(i)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(layers):
            fn_of_i(i)
    for _ in range(2):
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
    return ts[len(ts) // 2]


def attn_rows():
    D = 128
    shapes = [("13B world1 target verify", 40, 130066, 18, (6, 7, 8, 12, 13)),
              ("13B world1 retrieval verify", 40, 12305, 17, (6, 7, 8, 12, 13)),
              ("7B world1 g16 target verify", 32, 130066, 18, (8, 16)),
              ("7B world1 g16 retrieval verify", 32, 12305, 17, (8, 16)),
              ("7B TP2 rank target verify", 16, 130066, 17, (16, 8, 32)),
              ("7B TP2 rank retrieval verify", 16, 12305, 17, (16, 8, 32)),
              ("13B TP8 rank target verify", 5, 130066, 18, (51, 64, 102, 8)),
              ("13B TP8 rank retrieval verify", 5, 12305, 17, (51, 64, 102, 8))]
    for label, H, sk, sq, splits in shapes:
        per = 2 * sk * H * D * 2
        nrep = max(2, min(32, (1 << 30) // per + 1))
        g = torch.Generator(device=DEV).manual_seed(1)
        k = [torch.randn(H, sk, D, generator=g, device=DEV, dtype=torch.float16) for _ in range(nrep)]
        v = [torch.randn(H, sk, D, generator=g, device=DEV, dtype=torch.float16) for _ in range(nrep)]
        q = torch.randn(sq, H, D, generator=g, device=DEV, dtype=torch.float16)
        row = {"kind": "attn", "shape": label, "H": H, "sk": sk, "sq": sq, "MB": round(per / 1e6, 1),
               "default_nsplit": ops._pick_nsplit(H, sk)}
        for ns in (None,) + tuple(splits):
            us = chain_time(lambda i, ns=ns: ops.attn_decode(q, k[i % nrep], v[i % nrep], sk, 0.08837890625, nsplit=ns))
            row[f"nsplit_{ns or 'default'}_us"] = round(us, 2)
            row[f"nsplit_{ns or 'default'}_TBps"] = round(per / us / 1e6, 2)
        print(json.dumps(row), flush=True)
        del k, v


def score_rows():
    D, chunk = 128, 8
    for H, P in ((40, 130048), (32, 124928), (16, 130048), (5, 130048), (4, 124928)):
        C = P // chunk
        per = P * H * D * 2
        nrep = max(2, (1 << 30) // per + 1)
        g = torch.Generator(device=DEV).manual_seed(2)
        k = [torch.randn(H, P, D, generator=g, device=DEV, dtype=torch.float16) for _ in range(nrep)]
        q = torch.randn(H, D, generator=g, device=DEV, dtype=torch.float16)
        us = chain_time(lambda i: ops.retrieval_score(k[i % nrep], q, C, chunk), layers=16)
        print(json.dumps({"kind": "retrieval_score", "H": H, "P": P, "MB": round(per / 1e6, 1), "us": round(us, 2),
                          "TBps": round(per / us / 1e6, 2), "lib": os.environ.get("TRIFORCE_HIP_LIB", "default")}), flush=True)
        del k


if __name__ == "__main__":
    what = sys.argv[1:] or ["attn", "score"]
    if "score" in what:
        score_rows()
    if "attn" in what:
        attn_rows()
