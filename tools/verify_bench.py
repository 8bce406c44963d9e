"""Stage latencies of the decode loop's model calls at BASELINE configs[1] shapes without the 11 s prefill: random-init
7B weights, N(0,1) KV fill (the reference's own filler, cache.py:303-308), hipGraph replays as the loop issues them.
A/B of build variants / env switches:  TRIFORCE_HIP_LIB=... TRIFORCE_PREFETCH_NEXT=4 python tools/verify_bench.py <tag>
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "default"
    args = bench.parse(sys.argv[2:])
    dev = torch.device("cuda", 0)
    target, draft = bench.load_models(args, dev, "random", "random:1", "random:2")
    ge = bench.build_engine(args, dev, target, draft)
    eng = ge.engine
    g = torch.Generator(device=dev).manual_seed(1)
    for l in range(eng.kv_cache.layers):
        eng.kv_cache.k[l].normal_(generator=g)
        eng.kv_cache.v[l].normal_(generator=g)
    eng.graph_cache.k.normal_(generator=g)
    eng.graph_cache.v.normal_(generator=g)
    eng.draft_cache.k.normal_(generator=g)
    eng.draft_cache.v.normal_(generator=g)
    eng.kv_cache.seq_len = args.prefill
    out = {"tag": tag, "lib": os.environ.get("TRIFORCE_HIP_LIB", "default"),
           "prefetch_next": os.environ.get("TRIFORCE_PREFETCH_NEXT", "0")}
    reps = []
    for _ in range(3):
        reps.append(bench.stage_latencies(ge, args, dev))
    for k in reps[0]:
        out[k] = sorted(r[k] for r in reps)[1]                       # median of 3
    tok = torch.tensor([[100]], device=dev)
    S = eng.kv_cache.seq_len

    def ar():
        ge.decode_step(tok)
        eng.kv_cache.seq_len = S
    out["ar_step_us"] = round(bench._timed(ar, 5), 1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
