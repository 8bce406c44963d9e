"""Split-count table for tf_attn_decode: times every (heads on this rank, key count, query tile count) shape the
BASELINE configs run — 7B / 13B at TP 1..8, retrieval budgets 4096 / 12288, full caches of 125K / 130K — over the
candidate split counts, cold-cache (the K/V replicas rotate so that a small shape is not served from the 256 MiB
Infinity Cache), and prints the best split per shape next to what tf_attn_decode_pick_nsplit chooses today.

    python tools/nsplit_sweep.py [tag]        -> gpurun_out/nsplit_sweep_<tag>.json   (TRIFORCE_HIP_LIB selects a variant build)
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from triforce_amd import hip, ops  # noqa: E402

DEV = "cuda:0"
SCALE = 0.08837890625


def time_shape(sq, sk, H, D, splits, budget_bytes=1 << 30):
    per = 2 * sk * H * D * 2
    reps = max(1, min(16, budget_bytes // per))
    g = torch.Generator(device=DEV).manual_seed(sk + H)
    ks = [torch.randn(H, sk, D, generator=g, device=DEV, dtype=torch.float16) for _ in range(reps)]
    vs = [torch.randn(H, sk, D, generator=g, device=DEV, dtype=torch.float16) for _ in range(reps)]
    q = torch.randn(sq, H, D, generator=g, device=DEV, dtype=torch.float16)
    out = {}
    iters = max(reps * 2, 12)
    for ns in splits:
        try:
            for i in range(3):
                ops.attn_decode(q, ks[i % reps], vs[i % reps], sk, SCALE, nsplit=ns)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(iters):
                ops.attn_decode(q, ks[i % reps], vs[i % reps], sk, SCALE, nsplit=ns)
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) / iters * 1e3
            out[ns] = round(us, 1)
        except Exception as ex:                                     # split count the kernel refuses for this shape
            out[ns] = None
    del ks, vs
    torch.cuda.empty_cache()
    return out, per


if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else "default"
    L = hip.lib()
    res = {"lib": os.environ.get("TRIFORCE_HIP_LIB", "default"), "shapes": []}
    splits = [2, 4, 8, 12, 16, 24, 32, 48, 64, 96, 128]
    for H in (4, 5, 8, 16, 20, 32, 40):
        for sk in (4103, 12305, 124936, 130066):
            for sq in (7, 17):
                t, per = time_shape(sq, sk, H, 128, splits)
                valid = {k: v for k, v in t.items() if v}
                best = min(valid, key=valid.get)
                pick = L.tf_attn_decode_pick_nsplit(H, sk)
                row = {"H": H, "sk": sk, "sq": sq, "us": t, "best": best, "best_us": valid[best],
                       "best_GBps": round(per / valid[best] / 1e3, 1), "pick": pick, "pick_us": t.get(pick)}
                res["shapes"].append(row)
                print(json.dumps(row), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"nsplit_sweep_{tag}.json"), "w") as f:
        json.dump(res, f, indent=1)
