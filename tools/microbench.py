"""Per-kernel timings on the GPU box (HIP events on torch's current stream, where the C-ABI kernels are
enqueued).  Writes gpurun_out/microbench.json.  Usage: python tools/microbench.py [--quick]"""
import json
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from triforce_amd import ops  # noqa: E402

DEV = "cuda:0"
OUT = {}


def timeit(fn, warm=3, iters=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3          # us


def section(name):
    def deco(f):
        try:
            f()
        except Exception:
            OUT[name] = {"error": traceback.format_exc()[-800:]}
            print(name, "FAILED", OUT[name]["error"], flush=True)
        return f
    return deco


@section("attn")
def _attn():
    g = torch.Generator(device=DEV).manual_seed(0)
    for (sq, sk, H, D, tag) in [(8, 124935, 32, 128, "target_verify_cfg2"), (7, 4103, 32, 128, "retrieval_verify_cfg2"),
                                (1, 124929, 32, 128, "ar_decode_cfg2"), (18, 130066, 16, 128, "target_verify_cfg4_rank"),
                                (17, 12305, 16, 128, "retrieval_verify_cfg4_rank")]:
        k = torch.randn(H, sk, D, generator=g, device=DEV, dtype=torch.float16)
        v = torch.randn(H, sk, D, generator=g, device=DEV, dtype=torch.float16)
        q = torch.randn(sq, H, D, generator=g, device=DEV, dtype=torch.float16)
        scale = 0.08837890625
        res = {}
        default = None
        for ns in [None, 8, 16, 32, 64, 128]:
            try:
                us = timeit(lambda: ops.attn_decode(q, k, v, sk, scale, nsplit=ns))
            except Exception as ex:
                res[str(ns)] = str(ex)[:100]
                continue
            gbs = 2 * sk * H * D * 2 / us / 1e3
            res[str(ns)] = {"us": round(us, 1), "GBps": round(gbs, 1)}
            if ns is None:
                default = res[str(ns)]
        OUT["attn_" + tag] = res
        print("attn", tag, json.dumps(res), flush=True)
        del k, v


@section("retrieval")
def _retr():
    g = torch.Generator(device=DEV).manual_seed(1)
    H, D, P, B, chunk = 32, 128, 124928, 4096, 8
    k = torch.randn(H, P, D, generator=g, device=DEV, dtype=torch.float16)
    v = torch.randn(H, P, D, generator=g, device=DEV, dtype=torch.float16)
    q = torch.randn(H, D, generator=g, device=DEV, dtype=torch.float16)
    kr = torch.zeros(H, B + 7, D, device=DEV, dtype=torch.float16)
    vr = torch.zeros(H, B + 7, D, device=DEV, dtype=torch.float16)
    C, sets = P // chunk, B // chunk
    t_score = timeit(lambda: ops.retrieval_score(k, q, C, chunk))
    scores = ops.retrieval_score(k, q, C, chunk)
    t_topk = timeit(lambda: ops.retrieval_topk(scores, sets))
    idx = ops.retrieval_topk(scores, sets)
    t_gather = timeit(lambda: ops.retrieval_gather(k, v, idx, kr, vr, chunk))
    score_bytes, gather_bytes = P * H * D * 2, 4 * B * H * D * 2
    OUT["retrieval_cfg2_layer"] = {
        "score_us": round(t_score, 1), "score_GBps": round(score_bytes / t_score / 1e3, 1),
        "topk_us": round(t_topk, 1), "gather_us": round(t_gather, 1),
        "gather_GBps": round(gather_bytes / t_gather / 1e3, 1),
        "build_GBps": round((score_bytes + gather_bytes) / (t_score + t_topk + t_gather) / 1e3, 1)}
    print("retrieval", json.dumps(OUT["retrieval_cfg2_layer"]), flush=True)


@section("gemm")
def _gemm():
    res = {}
    for (name, N, K) in [("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008),
                         ("lm_head", 32000, 4096)]:
        w = torch.randn(N, K, device=DEV, dtype=torch.float16) * 0.02
        for M in [1, 7, 8, 128]:
            x = torch.randn(M, K, device=DEV, dtype=torch.float16)
            us = timeit(lambda: torch.nn.functional.linear(x, w))
            res[f"{name}_M{M}"] = {"us": round(us, 1), "GBps": round(N * K * 2 / us / 1e3, 1)}
        del w
    OUT["gemm_torch"] = res
    print("gemm", json.dumps(res), flush=True)


@section("skinny_gemm")
def _sg():
    res = {}
    for (name, N, K) in [("qkv", 12288, 4096), ("o", 4096, 4096), ("down", 4096, 11008), ("lm_head", 32000, 4096),
                         ("draft_qkv", 2304, 768), ("draft_head", 32000, 768)]:
        w = torch.randn(N, K, device=DEV, dtype=torch.float16) * 0.02
        pl = ops.PackedLinear(w)
        for M in [1, 8, 18]:
            x = torch.randn(M, K, device=DEV, dtype=torch.float16)
            us = timeit(lambda: ops.linear(x, pl))
            ref = timeit(lambda: torch.nn.functional.linear(x, w))
            res[f"{name}_M{M}"] = {"us": round(us, 1), "GBps": round(N * K * 2 / us / 1e3, 1), "hipblaslt_us": round(ref, 1)}
        del w, pl
    wgu = torch.randn(22016, 4096, device=DEV, dtype=torch.float16) * 0.02
    pl = ops.PackedLinear(wgu, split=2)
    for M in [1, 8, 18]:
        x = torch.randn(M, 4096, device=DEV, dtype=torch.float16)
        us = timeit(lambda: ops.mlp_act(x, pl))
        ref = timeit(lambda: ops.silu_mul(torch.nn.functional.linear(x, wgu)))
        res[f"gate_up_swiglu_M{M}"] = {"us": round(us, 1), "GBps": round(22016 * 4096 * 2 / us / 1e3, 1), "hipblaslt_plus_silu_us": round(ref, 1)}
    OUT["skinny_gemm"] = res
    print("skinny_gemm", json.dumps(res), flush=True)


@section("glue")
def _glue():
    res = {}
    x = torch.randn(8, 4096, device=DEV, dtype=torch.float16)
    r = torch.randn(8, 4096, device=DEV, dtype=torch.float16)
    w = torch.ones(4096, device=DEV, dtype=torch.float16)
    res["rmsnorm_8x4096_us"] = round(timeit(lambda: ops.rmsnorm(x, w, 1e-5, residual=r, sum_out=r)), 2)
    gu = torch.randn(8, 22016, device=DEV, dtype=torch.float16)
    res["silu_mul_8x11008_us"] = round(timeit(lambda: ops.silu_mul(gu)), 2)
    H, D, T = 32, 128, 4200
    qkv = torch.randn(8, 3 * H * D, device=DEV, dtype=torch.float16)
    cos = torch.randn(131072, D, device=DEV, dtype=torch.float16)
    pos = torch.arange(1000, 1008, device=DEV)
    kc = torch.zeros(H, T, D, device=DEV, dtype=torch.float16)
    vc = torch.zeros(H, T, D, device=DEV, dtype=torch.float16)
    res["rope_append_8rows_us"] = round(timeit(lambda: ops.rope_append(qkv, cos, cos, pos, kc, vc, 10, H, D)), 2)
    logits = torch.randn(8, 32000, device=DEV)
    from triforce_amd.utils.sampling import norm_logits
    res["norm_logits_8x32000_us"] = round(timeit(lambda: norm_logits(logits, 0.6, -1, 0.9)), 1)
    res["norm_logits_1x32000_us"] = round(timeit(lambda: norm_logits(logits[:1], 0.6, -1, 0.9)), 1)
    from triforce_amd.utils.sampling import top_k_top_p_filter
    res["norm_logits_torch_sort_8x32000_us"] = round(timeit(lambda: torch.softmax(top_k_top_p_filter(logits / 0.6, -1, 0.9), -1)), 1)
    p = torch.softmax(logits, -1)
    out = torch.zeros(4, dtype=torch.int64, device=DEV)
    toks = torch.randint(0, 32000, (7,), device=DEV)
    u = torch.rand(8, device=DEV)
    res["accept_chain_us"] = round(timeit(lambda: ops.accept_chain(p, p[:7].contiguous(), toks, u, 7, False, 2, out)), 1)
    res["sample_us"] = round(timeit(lambda: ops.sample_inverse_cdf(p[0], u[:1], out[:1])), 1)
    # draft attention shape
    Hd, Dd = 12, 64
    qd = torch.randn(7, Hd, Dd, device=DEV, dtype=torch.float16)
    kd = torch.randn(Hd, 259, Dd, device=DEV, dtype=torch.float16)
    cd = torch.randn(2048, Dd, device=DEV, dtype=torch.float16)
    res["draft_attn_7x259_us"] = round(timeit(lambda: ops.attn_rope_on_read(qd, kd, kd, cd, cd, 259, 0.125)), 1)
    OUT["glue"] = res
    print("glue", json.dumps(res), flush=True)


if __name__ == "__main__":
    print("device", torch.cuda.get_device_name(0), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "microbench.json"), "w") as f:
        json.dump(OUT, f, indent=1)
    print("wrote gpurun_out/microbench.json")
