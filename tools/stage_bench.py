"""Stage latencies of the BASELINE configs[1] decode step without the 25 s real prefill: synthetic N(0,1) KV prefix
(the reference's own filler, cache.py:303-308), then the three model calls of a step timed with HIP events.
Usage:  [TRIFORCE_FUSE=none|rope|all2|all] python tools/stage_bench.py [--prefill N] [--gamma G]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

sys.argv = [sys.argv[0]] + sys.argv[1:] + ["--prefill-mode", "synthetic", "--no-cpu-baseline"]
args = bench.parse()
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
from triforce_amd import ops  # noqa: E402
from triforce_amd.utils.decoding import TriForceRunner  # noqa: E402
from triforce_amd.utils.sampling import UniformSource  # noqa: E402

ge = bench.build_engine(args, device)
tcfg, _ = bench.target_config(args.target)
ids = torch.randint(3, tcfg.vocab_size, (1, args.prefill), generator=torch.Generator().manual_seed(0)).to(device)
run = TriForceRunner(bench._Tok(), ge, args.gamma, top_k=-1, top_p=args.top_p, temperature=args.temp,
                     rng=UniformSource(device, seed=0))
bench.do_prefill(run, ge, ids, "synthetic")
for _ in range(2):
    run.step()
torch.cuda.synchronize()
out = bench.stage_latencies(ge, args, device)
out["fuse_mode"] = ops.FUSE_MODE
print(json.dumps(out), flush=True)
