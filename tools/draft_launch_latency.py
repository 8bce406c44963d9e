"""From an IDLE queue, how long does one 68M draft step take end to end on the host clock — replayed as its hipGraph
(what the decode loop does after every accept record) against the same forward issued eagerly through the native entry
point (13 launches from one C call)?  The difference is launch latency: DESIGN 13.6 finds the hop between an accept kernel
and the next draft graph's first kernel at 12-61 us, process to process.

    python tools/draft_launch_latency.py > profiles/r04_draft_launch_latency.json
"""
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


@torch.inference_mode()
def main():
    args = bench.parse(sys.argv[1:])
    dev = torch.device("cuda", 0)
    target, draft = bench.load_models(args, dev, "random", "random:1", "random:2")
    ge = bench.build_engine(args, dev, target, draft)
    eng = ge.engine
    eng.draft_cache.k.normal_()
    eng.draft_cache.v.normal_()
    ge.tok_buf.fill_(100)
    out = {"what": "wall time of one 68M draft step from an idle queue: sync; t0; launch; sync; t1 (host clock, us)", "rows": {}}
    for n in (0, 3, 5):
        fn = ge.callables[n]
        ids = ge.tok_buf[:, :n + 1]
        kw = ge.sampling

        def graph():
            fn.graph.replay()

        def eager():
            eng.draft_run(input_ids=ids, gamma_offset=n, **kw)

        row = {}
        for name, f in (("graph_replay", graph), ("eager_native_call", eager), ("graph_replay_again", graph)):
            for _ in range(20):
                f()
            torch.cuda.synchronize()
            ts = []
            for _ in range(300):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                f()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e6)
            ts.sort()
            row[name] = {"median_us": round(statistics.median(ts), 1), "p10_us": round(ts[30], 1), "p90_us": round(ts[270], 1)}
        # device-side duration of the same graph, back to back (launch latency amortised): the floor of the numbers above
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(50):
            fn.graph.replay()
        e.record()
        torch.cuda.synchronize()
        row["graph_back_to_back_us"] = round(s.elapsed_time(e) * 1e3 / 50, 1)
        out["rows"][f"gamma_offset_{n}"] = row
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
