"""A/B of the split-KV attention's merge: two launches (split + merge kernel) vs one (last workgroup of a head merges).
Cold-cache chain: K/V replicas rotate through > 1 GB so every launch streams from HBM, hipGraph of 32 launches (one per
"layer") timed with events.  python tools/attn_merge_ab.py > profiles/<file>.jsonl"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from triforce_amd import ops  # noqa: E402

DEV = torch.device("cuda", 0)
SHAPES = [  # (label, H, sk, sq)
    ("7B retrieval verify (cfg2)", 32, 4103, 7), ("7B target verify (cfg2)", 32, 124935, 7),
    ("7B retrieval verify gamma16 (cfg3)", 32, 12305, 17), ("TP2 shard target verify (cfg3)", 16, 130066, 17),
    ("TP2 shard retrieval verify gamma16 (cfg3)", 16, 12305, 17), ("7B target verify gamma16 (cfg3 @ world 1)", 32, 130066, 18),
    ("TP8 shard retrieval verify", 4, 4103, 7), ("13B TP8 shard target verify (cfg4)", 5, 130066, 17),
    ("68M draft-sized", 12, 256, 1),
]


def time_chain(H, sk, sq, fused, layers=32, reps=20):
    D = 128 if H != 12 else 64
    per = 2 * sk * H * D * 2
    nrep = max(2, min(layers, (1 << 30) // per + 1))
    g = torch.Generator(device=DEV).manual_seed(1)
    k = [torch.randn(H, sk, D, generator=g, device=DEV, dtype=torch.float16) for _ in range(nrep)]
    v = [torch.randn(H, sk, D, generator=g, device=DEV, dtype=torch.float16) for _ in range(nrep)]
    q = torch.randn(sq, H, D, generator=g, device=DEV, dtype=torch.float16)
    ops.ATTN_FUSED_MERGE = fused
    scale = D ** -0.5

    def chain():
        out = None
        for i in range(layers):
            out = ops.attn_decode(q, k[i % nrep], v[i % nrep], sk, scale)
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


def accuracy(H, sk, sq, D=128):
    """How exact is the kernel's output?  Against attention accumulated in fp64 (on the device) and rounded once to
    fp16: share of output elements that differ, and the mean deviation in units of the exact value's fp16 spacing."""
    g = torch.Generator(device=DEV).manual_seed(7)
    k = torch.randn(H, sk, D, generator=g, device=DEV, dtype=torch.float16)
    v = torch.randn(H, sk, D, generator=g, device=DEV, dtype=torch.float16)
    q = torch.randn(sq, H, D, generator=g, device=DEV, dtype=torch.float16)
    scale = 0.08837890625
    got = ops.attn_decode(q, k, v, sk, scale).view(sq, H, D)
    s = torch.einsum("qhd,hkd->hqk", q.double(), k.double()) * scale
    qi, kj = torch.arange(sq, device=DEV).view(sq, 1), torch.arange(sk, device=DEV).view(1, sk)
    s = s.masked_fill(kj > (sk - sq + qi), float("-inf"))
    want = torch.einsum("hqk,hkd->qhd", torch.softmax(s, dim=-1), v.double())
    w16 = want.half()
    ulp = torch.pow(2.0, torch.floor(torch.log2(want.abs().clamp_min(2.0 ** -14))) - 10)
    return {"outputs_differing_from_exact": round(float((got != w16).float().mean()), 4),
            "mean_err_in_fp16_ulps": round(float(((got.double() - want).abs() / ulp).mean()), 4)}


def main():
    # --fused-only [tag]: the shipped (one-launch where <= 8 splits) form only, two passes — for A/B of kernel-library
    # variants (TRIFORCE_HIP_LIB=...); the tag names the variant in every row
    fused_only = len(sys.argv) > 1 and sys.argv[1] == "--fused-only"
    tag = sys.argv[2] if len(sys.argv) > 2 else os.environ.get("TRIFORCE_HIP_LIB", "default")
    for label, H, sk, sq in SHAPES:
        row = {"shape": label, "H": H, "sk": sk, "sq": sq, "nsplit": ops._pick_nsplit(H, sk)}
        if fused_only:
            row["lib"] = tag
            a, per = time_chain(H, sk, sq, True, reps=10)
            b, _ = time_chain(H, sk, sq, True, reps=10)
            row.update(us=round(min(a, b), 2), us_other_pass=round(max(a, b), 2), GBps=round(per / min(a, b) / 1e3, 1))
            if sk <= 13000 and H != 12:
                row.update(accuracy(H, sk, sq))
            print(json.dumps(row), flush=True)
            continue
        for name, fused in (("two_launch_us", False), ("one_launch_us", True), ("two_launch_us_again", False),
                            ("one_launch_us_again", True)):
            us, per = time_chain(H, sk, sq, fused)
            row[name] = round(us, 2)
        row["GBps_one_launch"] = round(per / min(row["one_launch_us"], row["one_launch_us_again"]) / 1e3, 1)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
