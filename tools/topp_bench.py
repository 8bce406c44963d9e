"""Temperature + top-p + softmax: one workgroup per row (tf_topp_probs) against every row over 16 / 8 workgroups of one launch
(tf_topp_probs_multi), 50 calls per hipGraph, HIP events.  One JSON line per (rows, kind).
    python tools/topp_bench.py [--out gpurun_out/topp_bench.jsonl]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from triforce_amd import ops  # noqa: E402

DEV = "cuda:0"


def rows_of(rows, V, kind):
    g = torch.Generator().manual_seed(7 + rows)
    lg = torch.randn(rows, V, generator=g) * 2.5
    lg = {"model-like": lg, "sharp": lg * 4, "flat": lg * 0.01}[kind].half().float()
    return lg.to(DEV)


def graph_us(fn, calls=50, reps=5):
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(calls):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (reps * calls) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--list-cap", type=int, default=0)
    ap.add_argument("--rowmax", type=int, default=1, help="1: row maximum from the whole row (default); 0: slice maxima through an in-launch edge")
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    from triforce_amd import hip
    hip.lib().tf_topp_multi_tune(3, a.rowmax)
    if a.list_cap:
        from triforce_amd import hip
        hip.lib().tf_topp_multi_tune(2, a.list_cap)
    lines = []
    V = 32000
    with torch.inference_mode():
        for kind in ("model-like", "sharp", "flat"):
            for rows in (1, 7, 8, 18):
                lg = rows_of(rows, V, kind)
                pm = torch.full((V // 16, 32), float("-inf"), dtype=torch.float32, device=DEV)
                pm[:, :rows] = lg.view(rows, V // 16, 16).max(-1).values.t()
                out = {}
                for name, multi, panel in (("one_workgroup_per_row_us", False, None), ("multi_us", True, None), ("multi_panel_max_us", True, pm)):
                    ops.TOPP_MULTI = multi
                    out[name] = round(graph_us(lambda: ops.topp_probs(lg, 0.6, 0.9, panel_max=panel)), 2)
                kept = int((ops.topp_probs(lg, 0.6, 0.9)[0] > 0).sum())
                line = {"what": "temperature + top-p + softmax, V = 32000, T 0.6 / top_p 0.9, 50 calls per hipGraph", "rows": rows, "kind": kind, "list_cap": a.list_cap or 16384, "rowmax_from_row": a.rowmax, "tag": a.tag,
                        "kept_in_row_0": kept, **out}
                lines.append(line)
                print(json.dumps(line), flush=True)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "a") as f:
            for ln in lines:
                f.write(json.dumps(ln) + "\n")


if __name__ == "__main__":
    main()
