#!/usr/bin/env python
"""bench.py — TriForce decode throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (N=1): BASELINE.json configs[1] — Llama2-7B-128K (YaRN) on-chip, prefill 124 928, retrieval
budget 4 096, chunk 8, gamma 6, T=0.6 / top_p=0.9 (README.md:49-55), random-init weights and random
prompt tokens (no checkpoints / datasets offline).  A *step* is one outer TriForce iteration
(utils/decoding.py:70-141): gamma-bounded Middle_Spec drafting (68M draft graphs + retrieval-verify
graph), one target verify over the full 125K-token KV cache, device-side accept/rollback and the
cache fix-ups.  The timed region starts after prefill + retrieval build + draft prefill, exactly
like the reference's time1/time2 (decoding.py:69,143), with device syncs added on both sides.

value = tokens emitted in the K timed steps / wall time (tokens/s, whole job).
roofline = the dominant kernel (split-KV target-verify attention, tf_attn_decode): algorithmic bytes per
launch 2*S*H*D*2 / mean launch duration from HIP events recorded on the launch stream
inside the timed region (every 8th launch is bracketed: an event record is a queue packet of its own, and
bracketing all 32 per target verify would add ~0.4 ms of gaps to the step being measured).
cpu_baseline = the CPU oracle (oracle/, kind "port") timed on this host for a bounded per-layer sample
of the same step, extrapolated to the step.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--target", default="llama-7B-128K", choices=["llama-7B-128K", "llama-13B-128K", "lwm-128K", "tiny"])
    ap.add_argument("--prefill", type=int, default=124928)
    ap.add_argument("--budget", type=int, default=4096)
    ap.add_argument("--chunk_size", type=int, default=8)
    ap.add_argument("--gamma", type=int, default=6)
    ap.add_argument("--temp", type=float, default=0.6)
    ap.add_argument("--top_p", type=float, default=0.9)
    ap.add_argument("--gen_cap", type=int, default=512, help="KV slack reserved for generated tokens")
    ap.add_argument("--prefill-mode", default="real", choices=["real", "synthetic"],
                    help="real = chunked prefill through the model; synthetic = N(0,1) KV fill "
                         "(the reference's own filler, DistributedSimpleCache.normal_, cache.py:303-308)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--rebuild-every", type=int, default=0,
                    help="re-select the retrieval cache every N target verifies (0 = the reference: once per prompt)")
    args = ap.parse_args()
    # a step emits at most gamma + 2 tokens: size the KV slack for the worst case of the requested run
    # (never more than the retrieval budget: its tail holds every generated token, reference cache.py:180-182)
    args.gen_cap = max(args.gen_cap, min((args.steps + args.warmup + 4) * (args.gamma + 2) + 64, args.budget))
    return args


def target_config(name):
    from triforce_amd.models import zoo
    return zoo.config(name), zoo.config("llama-68M")


class _Tok:
    eos_token_id = 2

    def decode(self, *a, **k):
        return ""


def build_engine(args, device):
    from triforce_amd.models.cache import FlashSimpleCache, RetrievalCache, StreamingLLMEvictionCache
    from triforce_amd.models.modeling_llama import LlamaForCausalLM
    from triforce_amd.models.modeling_llama_68m import LlamaForCausalLM as Draft68M
    from triforce_amd.utils.graph_infer import GraphInferenceEngine
    tcfg, dcfg = target_config(args.target)
    target = LlamaForCausalLM(tcfg, device).init_random(args.seed + 1)
    draft = Draft68M(dcfg, device).init_random(args.seed + 2)
    cache = FlashSimpleCache(target, args.prefill + args.gen_cap + 16)
    gcache = RetrievalCache(target, max_budget=args.budget, prefill=args.prefill, gamma=args.gamma,
                            chunk_size=args.chunk_size)
    dcache = StreamingLLMEvictionCache(draft, start_size=16, recent_size=256 - 16 - args.gamma, gamma=args.gamma)
    ge = GraphInferenceEngine(target, cache, gcache, draft, dcache)
    if args.no_graphs:
        ge.initialize_eager(args.gamma, probs=True, temperature=args.temp, top_p=args.top_p)
    else:
        ge.initialize_cuda_graph(args.gamma, probs=True, temperature=args.temp, top_p=args.top_p, verbose=False)
    return ge


def do_prefill(run, ge, input_ids, mode):
    """Everything the reference does before time1 (decoding.py:44-62)."""
    eng = ge.engine
    if mode == "real":
        run.prefill(input_ids)
        return
    eng.kv_cache.reset()
    eng.graph_cache.reset()
    eng.draft_cache.reset()
    P = input_ids.shape[1]
    g = torch.Generator(device=eng.model.device).manual_seed(1234)
    for l in range(eng.kv_cache.layers):             # synthetic prefix KV, layer by layer (bounded temporaries)
        eng.kv_cache.k[l, :, :P - 1].normal_(generator=g)
        eng.kv_cache.v[l, :, :P - 1].normal_(generator=g)
    eng.kv_cache.seq_len = P - 1
    logits = ge.inference(input_ids=input_ids[:, -1:])           # last prompt token: builds the retrieval cache
    ge.graph_draft_prefill(input_ids=input_ids)
    run.start(logits)


def stage_latencies(ge, args, device):
    """Per-stage latency (us) of the three model calls of a step, HIP events on the launch stream."""
    eng = ge.engine
    gamma = args.gamma

    def t(fn, n=5):
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3

    S = eng.kv_cache.seq_len
    ids = torch.full((1, gamma + 1), 100, dtype=torch.long, device=device)
    pos = torch.arange(S, S + gamma + 1, device=device).unsqueeze(0)
    out = {"draft_step_us": t(lambda: ge.graph_draft_inference(ids[:, :3], gamma_offset=2)),
           "retrieval_verify_us": t(lambda: ge.graph_verify(ids, pos))}

    def tv():
        ge.inference(ids)
        eng.kv_cache.seq_len = S                   # roll the probe back
    out["target_verify_us"] = t(tv, n=3)

    def ar():
        ge.engine.model(input_ids=ids[:, :1], kv_cache=eng.kv_cache, graph_cache=None)
        eng.kv_cache.seq_len = S
    out["ar_decode_step_us"] = t(ar, n=3)
    return {k: round(v, 1) for k, v in out.items()}


def cpu_baseline(args, tokens_per_step, inner_per_step):
    """CPU oracle ('port') on this host: one target layer at full cfg shape + lm_head + one retrieval-verify
    layer + one draft forward, extrapolated to a full step.  Bounded to ~10-30 s of CPU work."""
    from oracle import ref_ops as R
    tcfg, dcfg = target_config(args.target)
    H, D, hid, I, L, V = (tcfg.num_attention_heads, tcfg.head_dim, tcfg.hidden_size, tcfg.intermediate_size,
                          tcfg.num_hidden_layers, tcfg.vocab_size)
    q_t, q_r = args.gamma + 2, args.gamma + 1
    S, Rb = args.prefill + q_t, args.budget + args.gamma + 1
    torch.manual_seed(0)
    t0 = time.time()
    kv = torch.empty(S, H, D, dtype=torch.float16).normal_()
    w = {n: (torch.empty(o, i, dtype=torch.float16).normal_() * 0.02) for n, (o, i) in
         dict(qkv=(3 * hid, hid), o=(hid, hid), gu=(2 * I, hid), d=(hid, I), head=(V, hid)).items()}
    setup = time.time() - t0

    def tm(fn):
        t = time.time()
        fn()
        return time.time() - t

    def gemms(rows):
        x = torch.randn(rows, hid).half()
        h = R.rms_norm(x, torch.ones(hid, dtype=torch.float16), 1e-5)
        R.linear(h, w["qkv"])
        R.linear(h, w["o"])
        a = R.linear(h, w["gu"])
        R.linear(R.silu_mul(a[:, :I], a[:, I:]), w["d"])

    scale = R.softmax_scale_for(D)
    qt, qr = torch.randn(q_t, H, D).half(), torch.randn(q_r, H, D).half()
    gemms(q_r)                                                        # warm the CPU GEMM path
    t_attn_full = tm(lambda: R.attn_kvcache(qt, kv, kv, scale))
    t_attn_retr = tm(lambda: R.attn_kvcache(qr, kv[:Rb], kv[:Rb], scale))
    t_gemm_t, t_gemm_r = tm(lambda: gemms(q_t)), tm(lambda: gemms(q_r))
    t_head = tm(lambda: R.norm_logits(R.linear(torch.randn(q_r, hid).half(), w["head"]).float(), args.temp, -1, args.top_p))
    # draft forward (2 layers, hidden 768) ~ 87 MB of weights: time the lm_head-sized part + layers
    dw = (torch.empty(dcfg.vocab_size, dcfg.hidden_size, dtype=torch.float16).normal_() * 0.02)
    t_draft = tm(lambda: R.linear(torch.randn(4, dcfg.hidden_size).half(), dw)) * (87.0 / 49.0)
    target_fwd = L * (t_attn_full + t_gemm_t) + t_head
    retr_fwd = L * (t_attn_retr + t_gemm_r) + t_head
    step = target_fwd + inner_per_step * retr_fwd + (inner_per_step + 1) * t_draft
    return {"value": round(tokens_per_step / step, 4), "unit": "tokens/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": (f"oracle (torch CPU) timed for 1 of {L} target layers at full shape (attention q={q_t} x S={S} "
                       f"keys + the 4 GEMMs), lm_head+top-p, 1 retrieval-verify layer (R={Rb}) and the draft GEMM; "
                       f"step = target fwd + {inner_per_step:.2f} x retrieval fwd + {inner_per_step + 1:.2f} x draft, "
                       f"extrapolated x{L} layers; {setup + t_attn_full + t_attn_retr + t_gemm_t + t_gemm_r + t_head:.1f} s "
                       f"of CPU work; est. {step:.1f} s per step"),
            "step_seconds_est": round(step, 2)}


def projection(stages, gamma, overhead_us, pairs=((0.5, 0.9), (0.7, 0.9), (0.9, 0.95))):
    """NOT a measurement: tokens/s the measured stage latencies would give if the models agreed like trained ones do.
    Random-init weights make the 68M draft and the target disagree (acceptance ~1 %), which pins every step at the
    worst case (gamma inner iterations, ~1 token).  For a (draft->retrieval, retrieval->target) per-token acceptance
    pair the loop of utils/decoding.py:70-141,163-223 is simulated 20 000 times and priced with
    step = target_verify + k * (retrieval_verify + draft) + draft + measured per-step overhead."""
    import random
    rnd = random.Random(0)
    out = {}
    for a1, a2 in pairs:
        tok = t_us = 0.0
        for _ in range(20000):
            n = k = 0
            while n < gamma:                                  # Middle_Spec: +2 tokens on accept, +1 on reject
                k += 1
                n += 2 if rnd.random() < a1 else 1
            count = 0
            while count < n and rnd.random() < a2:            # accepted prefix, then the resample / bonus token
                count += 1
            tok += count + 1
            t_us += stages["target_verify_us"] + k * (stages["retrieval_verify_us"] + stages["draft_step_us"]) \
                + stages["draft_step_us"] + overhead_us
        out[f"draft_acc={a1},retrieval_acc={a2}"] = {"tokens_per_s": round(tok / t_us * 1e6, 1),
                                                     "tokens_per_step": round(tok / 20000, 2)}
    return out


def pmc_traffic(alg_bytes, H, D):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE and
    WRITE_SIZE collected in separate runs, KiB units, FETCH_SIZE doubled for gfx950 — MI355X_MICROARCH.md §HBM).
    PMC counters cannot be read from inside this process; the latest profiles/*pmc_attn*.json is used when it
    was measured on the same shape (H, D, key count within 0.1%), else traffic stays null."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_attn*.json"))):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        sh = j.get("shape", {})
        if sh.get("H") == H and sh.get("D") == D and abs(j["algorithmic_bytes_per_launch"] / alg_bytes - 1) < 1e-3:
            best = (f, j)
    if best is None:
        return {"traffic": None}
    f, j = best
    return {"traffic": j["traffic_bytes_per_launch"], "traffic_unit": "bytes/launch",
            "traffic_source": os.path.relpath(f, ROOT), "traffic_over_algorithmic": j["traffic_over_algorithmic"]}


def attn_roofline(timer, retrieval_rows, H, D):
    """Live roofline of the dominant kernel — target-verify attention over the full KV (the sampled launches with
    more keys than the retrieval cache holds).  H = heads on THIS rank.  HIP events on the launch stream."""
    from triforce_amd import ops
    full = [(a.elapsed_time(b) * 1e-3, sk) for (a, b, sk, _, _) in timer if sk > retrieval_rows]
    if not full:
        return None
    dur = sum(d for d, _ in full) / len(full)
    byts = sum(2 * sk * H * D * 2 for _, sk in full) / len(full)
    achieved = byts / dur / 1e9
    roof = {"bound": "hbm", "kernel": "attn_split_kernel<128,1> (+merge) via tf_attn_decode",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
            "launches": len(full), "launches_sampled_every": ops.ATTN_TIMER_EVERY,
            "avg_launch_us": round(dur * 1e6, 1), "algorithmic_bytes_per_launch": int(byts)}
    roof.update(pmc_traffic(byts, H, D))
    return roof


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1 or os.environ.get("TRIFORCE_BENCH_FORCE_TP") == "1":
        from bench_tp import run_tp                  # tensor-parallel decode (heads sharded, RCCL all-reduce)
        return run_tp(args, rank, world, local)

    from triforce_amd import ops
    from triforce_amd.utils.decoding import TriForceRunner
    from triforce_amd.utils.sampling import UniformSource

    t_setup = time.time()
    ge = build_engine(args, device)
    tcfg, _ = target_config(args.target)
    gen = torch.Generator().manual_seed(args.seed)
    input_ids = torch.randint(3, tcfg.vocab_size, (1, args.prefill), generator=gen).to(device)
    run = TriForceRunner(_Tok(), ge, args.gamma, top_k=-1, top_p=args.top_p, temperature=args.temp,
                         rng=UniformSource(device, seed=args.seed), rebuild_every=args.rebuild_every)
    t0 = time.time()
    do_prefill(run, ge, input_ids, args.prefill_mode)
    torch.cuda.synchronize()
    t_prefill = time.time() - t0

    for _ in range(args.warmup):
        run.step()
    torch.cuda.synchronize()
    n0, steps0, acc0, dr0 = run.n, len(run.counts), run.accepted_count, run.draft_count
    inner0 = run.inner_iters
    ops.ATTN_TIMER = []
    torch.cuda.synchronize()
    t1 = time.time()
    for _ in range(args.steps):
        run.step()
    torch.cuda.synchronize()
    t2 = time.time()
    timer, ops.ATTN_TIMER = ops.ATTN_TIMER, None
    seconds = t2 - t1
    tokens = run.n - n0
    accepted, drafted = run.accepted_count - acc0, run.draft_count - dr0
    value = tokens / seconds

    roof = attn_roofline(timer, args.budget + args.gamma + 1, tcfg.num_attention_heads, tcfg.head_dim)

    stages = stage_latencies(ge, args, device)
    inner_per_step = (run.inner_iters - inner0) / max(args.steps, 1)
    result = {
        "metric": "decode tokens/sec + avg accepted len, Llama-7B-128K @124K ctx",
        "value": round(value, 3), "unit": "tokens/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(seconds / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: {tcfg._name_or_path} on-chip TriForce decode, prefill "
                               f"{args.prefill}, budget {args.budget}, chunk {args.chunk_size}, gamma {args.gamma}, "
                               f"T={args.temp}, top_p={args.top_p}, 1xMI355X",
                   "prefill_mode": args.prefill_mode, "weights": "random-init N(0,0.02) fp16",
                   "hipgraphs": not args.no_graphs, "retrieval_rebuild_every": args.rebuild_every},
        "avg_accepted_len": round(accepted / max(drafted, 1) * args.gamma, 4),
        "acceptance_rate": round(accepted / max(drafted, 1), 4),
        "tokens": tokens, "tokens_per_step": round(tokens / args.steps, 3),
        "drafted_per_step": round(drafted / max(args.steps, 1), 3),
        "inner_iterations_per_step": round(inner_per_step, 3),
        "stage_latency_us": stages,
        "ar_baseline_tokens_per_s": round(1e6 / stages["ar_decode_step_us"], 2),
        "prefill_seconds": round(t_prefill, 2), "setup_seconds": round(t0 - t_setup, 2),
        "kv_seq_len": ge.engine.kv_cache.seq_len,
        "roofline": roof,
    }
    modelled = stages["target_verify_us"] + inner_per_step * (stages["retrieval_verify_us"] + stages["draft_step_us"]) \
        + stages["draft_step_us"]
    overhead_us = max(0.0, seconds / args.steps * 1e6 - modelled)
    result["step_overhead_us"] = round(overhead_us, 1)          # accept kernels, cache fix-ups, host round trips
    result["projection_not_measured"] = projection(stages, args.gamma, overhead_us)
    if not args.no_cpu_baseline:
        inner_iters = (run.inner_iters - inner0) / max(args.steps, 1)   # 68M drafts + retrieval verifies per step
        try:
            result["cpu_baseline"] = cpu_baseline(args, tokens / args.steps, inner_iters)
        except Exception as ex:                                    # host too small for the sample: report, don't fake
            result["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": torch.get_num_threads(),
                                      "kind": "port", "sample": f"failed: {type(ex).__name__}: {ex}"}
    print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
