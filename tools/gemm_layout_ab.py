"""Round 4, verdict item 1a: the skinny decode GEMMs at 8 ... 32 activation rows, 7B and 13B widths, by activation layout
(row-major vs k-octet-major, include/triforce_hip.h tf_skinny_gemm_act) and by panels per wave (tf_sg_tune).  Cold-cache
hipGraph chains as in tools/gemm_rows_ab.py; every variant's output is first checked against the row-major one-panel form
(bit-identical for the same wave count) and against a torch fp32 product.

    python tools/gemm_layout_ab.py > profiles/r04_gemm_layout_ab.jsonl
"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from triforce_amd import hip, ops  # noqa: E402

DEV = "cuda:0"
L = hip.lib()


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fns, iters=40):
    for f in fns:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters):
            fns[i % len(fns)]()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters * 1e3)
    return best


def call(pl, split, xbuf, xs, ybuf, ys, M, resid=None, rs=(8, 8)):
    """One launch through the layout-explicit entry points."""
    if split == 2:
        I = pl.N // 2
        hip.check(L.tf_skinny_gemm_swiglu_act(_p(pl.parts[0]), _p(pl.parts[1]), _p(xbuf), xs[0], xs[1], None, 0.0, None,
                                              _p(ybuf), ys[0], ys[1], M, I, pl.K, stream()), "swiglu_act")
    else:
        hip.check(L.tf_skinny_gemm_act(_p(pl.wp), _p(xbuf), xs[0], xs[1], None, 0.0, None, _p(resid), rs[0], rs[1], None,
                                       _p(ybuf), ys[0], ys[1], M, pl.N, pl.K, 0, stream()), "gemm_act")


# launch rules as tf_sg_tune settings: key 0 rows from which two panels per wave run (33 never), 1 its waves, 2 smallest halved
# grid that takes it, 4 K-splits ACROSS workgroups (1 = one workgroup per panel, > 1 forced, 0 = the shipped rule)
RULES = {"p1": {0: 33, 1: 8, 2: 256, 4: 1}, "p2w8": {0: 1, 1: 8, 2: 256, 4: 1}, "p2w4": {0: 1, 1: 4, 2: 256, 4: 1},
         "p1ks2": {0: 33, 1: 8, 2: 256, 4: 2}, "p1ks3": {0: 33, 1: 8, 2: 256, 4: 3}}
if os.environ.get("GEMM_RULES"):
    RULES = {k: RULES[k] for k in os.environ["GEMM_RULES"].split(",")}


def set_rule(name):
    for key, val in RULES[name].items():
        L.tf_sg_tune(key, val)


def main():
    rows_list = [int(v) for v in os.environ.get("GEMM_ROWS", "8,16,17,24,32").split(",")]
    only = set(os.environ.get("GEMM_ONLY", "").split(",")) - {""}            # e.g. "13B:o,13B:down"
    for model, hid, inter in (("7B", 4096, 11008), ("13B", 5120, 13824)):
        for name, N, K, split in (("qkv", 3 * hid, hid, 1), ("o", hid, hid, 1), ("gate_up", 2 * inter, hid, 2),
                                  ("down", hid, inter, 1)):
            if only and f"{model}:{name}" not in only:
                continue
            copies = max(2, int(700e6 // (N * K * 2)) + 1)
            pls = [ops.PackedLinear(torch.randn(N, K, device=DEV, dtype=torch.float16) * 0.02, split=split)
                   for _ in range(copies)]
            Nout = N // split
            row = {"model": model, "gemm": name, "N": N, "K": K, "MB": round(N * K * 2 / 1e6, 1)}
            for M in rows_list:
                x = torch.randn(M, K, device=DEV, dtype=torch.float16)
                xp = ops.pack_act(x)
                y_rm = torch.zeros(M, Nout, device=DEV, dtype=torch.float16)
                y_pk = torch.zeros(Nout // 8, M, 8, device=DEV, dtype=torch.float16)
                lay = {"rm": (x, (K, 8), y_rm, (Nout, 8)), "pk": (xp, (8, 8 * M), y_pk, (8, 8 * M))}
                # reference: torch fp32 product of the first weight copy
                w0 = pls[0].w.float()
                if split == 2:
                    g_, u_ = (x.float() @ w0[:Nout].T).half().float(), (x.float() @ w0[Nout:].T).half()
                    want = (torch.nn.functional.silu(g_).half() * u_)
                else:
                    want = (x.float() @ w0.T).half()
                base = None
                for rule in RULES:
                    set_rule(rule)
                    for lname, (xb, xs, yb, ys) in lay.items():
                        yb.zero_()
                        call(pls[0], split, xb, xs, yb, ys, M)
                        torch.cuda.synchronize()
                        got = yb if lname == "rm" else ops.unpack_act(yb, M)
                        err = float((got.float() - want.float()).abs().max())
                        tol = 4e-3 * max(1.0, float(want.float().abs().max()))
                        ok = err <= tol and bool(torch.isfinite(got).all())
                        if lname == "rm":
                            base = got.clone()
                        if RULES[rule].get(4, 0) > 1 and lname == "rm":
                            base = got.clone()                       # split-K re-associates the K sum: compare layouts within the rule
                        same = bool(torch.equal(got, base))
                        us = timeit([(lambda p=p: call(p, split, xb, xs, yb, ys, M)) for p in pls])
                        row[f"M{M}_{lname}_{rule}_us"] = round(us, 2)
                        if not (ok and same):
                            row[f"M{M}_{lname}_{rule}_BAD"] = {"err": err, "tol": tol, "same_as_rm": same}
            L.tf_sg_tune(0, 1), L.tf_sg_tune(1, 4), L.tf_sg_tune(2, 420), L.tf_sg_tune(4, 0)      # the shipped rule
            print(json.dumps(row), flush=True)
            del pls


if __name__ == "__main__":
    main()
