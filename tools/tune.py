"""Cold-cache A/B of compile-time kernel knobs.  Each GEMM shape cycles through enough distinct weight copies
(> 600 MB) that the 256 MiB Infinity Cache cannot serve re-reads — the in-situ condition of a 13 GB forward.
Usage:  TRIFORCE_HIP_LIB=triforce_amd/lib/libtriforce_hip_<variant>.so python tools/tune.py <tag>"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from triforce_amd import ops  # noqa: E402

DEV = "cuda:0"
tag = sys.argv[1] if len(sys.argv) > 1 else "default"
only_gemm = len(sys.argv) > 2 and sys.argv[2] == "gemm"


def timeit(fns, iters=40):
    n = len(fns)
    for i in range(n):
        fns[i]()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()                      # graph replay: no host launch floor in the numbers
    with torch.cuda.graph(g):
        for i in range(iters):
            fns[i % n]()
    g.replay()
    torch.cuda.synchronize()
    s.record()
    g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


res = {"tag": tag}
for (name, N, K, M) in [("qkv", 12288, 4096, 8), ("o", 4096, 4096, 8), ("down", 4096, 11008, 8), ("lm_head", 32000, 4096, 8)]:
    copies = max(2, int(700e6 // (N * K * 2)) + 1)
    pls = [ops.PackedLinear(torch.randn(N, K, device=DEV, dtype=torch.float16) * 0.02) for _ in range(copies)]
    x = torch.randn(M, K, device=DEV, dtype=torch.float16)
    us = timeit([(lambda p=p: ops.linear(x, p)) for p in pls])
    res[name] = {"us": round(us, 2), "GBps": round(N * K * 2 / us / 1e3, 1)}
    ws = [p.w for p in pls]
    us = timeit([(lambda w=w: torch.nn.functional.linear(x, w)) for w in ws])
    res[name]["hipblaslt_us"] = round(us, 2)
    del pls, ws
pls = [ops.PackedLinear(torch.randn(22016, 4096, device=DEV, dtype=torch.float16) * 0.02, split=2) for _ in range(4)]
x = torch.randn(8, 4096, device=DEV, dtype=torch.float16)
us = timeit([(lambda p=p: ops.mlp_act(x, p)) for p in pls])
res["gate_up_swiglu"] = {"us": round(us, 2), "GBps": round(22016 * 4096 * 2 / us / 1e3, 1)}
del pls
g = torch.Generator(device=DEV).manual_seed(0)
for (sq, sk, H, tagk) in ([] if only_gemm else [(8, 124936, 32, "attn_target"), (7, 4103, 32, "attn_retrieval")]):
    kvs = [(torch.randn(H, sk, 128, generator=g, device=DEV, dtype=torch.float16),
            torch.randn(H, sk, 128, generator=g, device=DEV, dtype=torch.float16)) for _ in range(2 if sk > 10000 else 12)]
    q = torch.randn(sq, H, 128, generator=g, device=DEV, dtype=torch.float16)
    for ns in ([None, 16, 64, 128] if sk > 10000 else [None, 8, 16, 32]):
        us = timeit([(lambda kv=kv: ops.attn_decode(q, kv[0], kv[1], sk, 0.08837890625, nsplit=ns)) for kv in kvs], iters=24)
        res[f"{tagk}_ns{ns}"] = {"us": round(us, 2), "GBps": round(2 * sk * H * 128 * 2 / us / 1e3, 1)}
    del kvs
if not only_gemm:
    # Sequoia verify: 512 tree nodes (4 slabs of 128 rows) over a 124 928-token prefix; random lower-triangular tree mask
    T, P, H = 512, 124928, 32
    kvs = [(torch.randn(H, P + T, 128, generator=g, device=DEV, dtype=torch.float16),
            torch.randn(H, P + T, 128, generator=g, device=DEV, dtype=torch.float16)) for _ in range(2)]
    q = torch.randn(T, H, 128, generator=g, device=DEV, dtype=torch.float16)
    vis = torch.tril(torch.rand(T, T, generator=g, device=DEV) < 0.2) | torch.eye(T, dtype=torch.bool, device=DEV)
    bits = ops.pack_tree_mask(vis)
    us = timeit([(lambda kv=kv: ops.attn_tree(q, kv[0], kv[1], P + T, 0.08838834764831845, bits, P)) for kv in kvs], iters=8)
    res["attn_tree_verify_512"] = {"us": round(us, 2), "us_per_slab": round(us / 4, 2)}
    del kvs
print(json.dumps(res), flush=True)
