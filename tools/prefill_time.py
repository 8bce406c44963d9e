"""Wall time of the 124 928-token target prefill (7B shapes, random weights) for one TRIFORCE_PREFILL_CHUNK setting.
    TRIFORCE_PREFILL_CHUNK=2048 python tools/prefill_time.py"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    args = bench.parse(["--no-graphs"] + sys.argv[1:])
    dev = torch.device("cuda", 0)
    target, draft = bench.load_models(args, dev, "random", "random:1", "random:2")
    ge = bench.build_engine(args, dev, target, draft)
    ids = torch.randint(3, 32000, (1, args.prefill), generator=torch.Generator().manual_seed(0)).to(dev)
    from triforce_amd.utils import graph_infer
    out = {"prefill_chunk": graph_infer.PREFILL_CHUNK, "prefill": args.prefill}
    ts = []
    for _ in range(2):
        ge.engine.kv_cache.reset()
        torch.cuda.synchronize()
        t0 = time.time()
        ge.inference(input_ids=ids[:, :-1])
        torch.cuda.synchronize()
        ts.append(time.time() - t0)
    out["target_prefill_seconds"] = [round(t, 3) for t in ts]
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
