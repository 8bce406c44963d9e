"""A/B of the chunked prefill's attention: one tf_attn_block launch per 128 rows vs tf_attn_prefill (one launch per
chunk, row blocks sharing the KV stream through L2).  7B shapes (32 heads x 128), chunks of 1024 rows at several cache
lengths; TF/s counts the causal 4 * rows * keys * H * D flops.  python tools/prefill_attn_ab.py > profiles/<file>.jsonl"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from triforce_amd import ops  # noqa: E402

DEV = torch.device("cuda", 0)


def bench(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    H, D = 32, 128
    chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    g = torch.Generator(device=DEV).manual_seed(1)
    cap = 124928
    k = torch.randn(H, cap, D, generator=g, device=DEV, dtype=torch.float16)
    v = torch.randn(H, cap, D, generator=g, device=DEV, dtype=torch.float16)
    q = torch.randn(chunk, H, D, generator=g, device=DEV, dtype=torch.float16)
    scale = D ** -0.5
    for sk in (chunk, 8192, 32768, 65536, 124928):
        flops = 4.0 * H * D * (chunk * (sk - chunk) + chunk * (chunk + 1) / 2)
        row = {"chunk_rows": chunk, "sk": sk, "nsplit_one_launch": ops.hip.lib().tf_attn_prefill_pick_nsplit(H, chunk, sk)}
        for name, flag in (("per_block_us", False), ("one_launch_us", True), ("per_block_us_again", False),
                           ("one_launch_us_again", True)):
            ops.ATTN_PREFILL_ONE_LAUNCH = flag
            row[name] = round(bench(lambda: ops.attn_prefill(q, k, v, sk, scale), 4 if sk > 40000 else 10), 1)
        row["TFps_per_block"] = round(flops / min(row["per_block_us"], row["per_block_us_again"]) / 1e6, 1)
        row["TFps_one_launch"] = round(flops / min(row["one_launch_us"], row["one_launch_us_again"]) / 1e6, 1)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
