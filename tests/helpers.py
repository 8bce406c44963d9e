"""Shared builders for the parity tests (oracle side and product side from the same seeds)."""
import os

import torch

from oracle import ref_model as M
from oracle import specs

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def note(msg):
    """A measured parity figure: printed (visible with -s / on failure) AND appended to gpurun_out/parity_notes.txt, which
    tools/gpu_validate.sh copies to profiles/ — so the measured max / mean deviations behind every tolerance are on
    record, not just the pass / fail bit."""
    line = f"[parity] {msg}"
    print(line, flush=True)
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_notes.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


def record(check, used, **info):
    """One row of the tolerance table (profiles/r06_parity_table.md): which test, which kind of check, how
    much of its bound the measured deviation used (1.0 = at the bound), and the raw figures.  Appended to
    gpurun_out/parity_table.jsonl; tools/parity_table.py folds the file into the table."""
    import json
    row = {"test": os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], "check": check, "used": round(float(used), 5)}
    row.update({k: (round(float(v), 9) if isinstance(v, float) else v) for k, v in info.items()})
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_table.jsonl"), "a") as f:
            f.write(json.dumps(row) + "\n")
    except OSError:
        pass


def bound(what, value, limit):
    """`value < limit`, with the pair put on record (a bare `assert x < 2e-2` leaves no trace of how close x came)."""
    record("bound", float(value) / float(limit), what=what, value=float(value), limit=float(limit))
    return float(value) < float(limit)


def close(got, want, what="", **kw):
    """torch.testing.assert_close with the measured deviation put on record: `used` = max |d| / (atol + rtol |want|)."""
    atol, rtol = kw.get("atol"), kw.get("rtol")
    if atol is not None and rtol is not None and got.shape == want.shape and got.numel():
        g, w = got.detach().float().cpu(), want.detach().float().cpu()
        d = (g - w).abs()
        lim = atol + rtol * w.abs()
        ok = lim > 0
        used = float((d[ok] / lim[ok]).max()) if bool(ok.any()) else (0.0 if float(d.max()) == 0.0 else float("inf"))
        record("assert_close", used, what=what, max_abs=float(d.max()), mean_abs=float(d.mean()), atol=float(atol), rtol=float(rtol),
               max_ref=float(w.abs().max()))
    torch.testing.assert_close(got, want, **kw)


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, f"{name}.pt"), weights_only=False)


class FakeTokenizer:
    eos_token_id = 2

    def decode(self, ids, **kw):
        return ""


def build_oracle(g, temperature=None, top_p=None):
    tsd = specs.random_state_dict(g["tcfg"], g["tseed"], head_std=g.get("head_std", 0.05))
    dsd = specs.random_state_dict(g["dcfg"], g["dseed"], head_std=g.get("head_std", 0.05))
    recent = 256 - 16 - g["gamma"]
    ot, od = M.OracleTarget(g["tcfg"], tsd), M.OracleDraft(g["dcfg"], dsd)
    okv = M.FullCache(g["tcfg"], g["prefill"] + g["gen_len"] + 16)
    ogc = M.RetrievalCacheO(g["tcfg"], g["budget"], g["prefill"], g["chunk"], g["gamma"])
    odc = M.StreamingCacheO(g["dcfg"], gamma=g["gamma"], start_size=16, recent_size=recent)
    eng = M.OracleEngine(ot, okv, ogc, od, odc, g["temperature"] if temperature is None else temperature,
                         g["top_p"] if top_p is None else top_p)
    return eng, tsd, dsd


def build_oracle_tp(g, temperature, top_p):
    """Oracle engine configured like the reference's TP path (tests/golden/tp_chain.pt): draft prefill in 128-token
    blocks, draft sampled at 0.6 / 0.9 whatever the target's settings, KV capacity prefill + gen_len + 32."""
    tsd = specs.random_state_dict(g["tcfg"], g["tseed"], head_std=g["head_std"])
    dsd = specs.random_state_dict(g["dcfg"], g["dseed"], head_std=g["head_std"])
    gamma = g["gamma"]
    eng = M.OracleEngine(M.OracleTarget(g["tcfg"], tsd), M.FullCache(g["tcfg"], g["prefill"] + g["gen_len"] + 32),
                         M.RetrievalCacheO(g["tcfg"], g["budget"], g["prefill"], g["chunk"], gamma),
                         M.OracleDraft(g["dcfg"], dsd),
                         M.StreamingCacheO(g["dcfg"], gamma=gamma, start_size=16, recent_size=256 - 16 - gamma),
                         temperature, top_p, draft_chunk=128, draft_temperature=0.6, draft_top_p=0.9)
    return eng, tsd, dsd


def build_product(g, device, tsd=None, dsd=None, temperature=None, top_p=None, graphs=False, target_graph=True):
    """The product engine (triforce_amd) from the same seeded weights."""
    from triforce_amd.models.cache import FlashSimpleCache, RetrievalCache, StreamingLLMEvictionCache
    from triforce_amd.models.config_yarn import LlamaConfig
    from triforce_amd.models.modeling_llama import LlamaForCausalLM
    from triforce_amd.models.modeling_llama_68m import LlamaForCausalLM as LlamaForCausalLM_68M
    from triforce_amd.utils.graph_infer import GraphInferenceEngine
    if tsd is None:
        tsd = specs.random_state_dict(g["tcfg"], g["tseed"], head_std=g.get("head_std", 0.05))
        dsd = specs.random_state_dict(g["dcfg"], g["dseed"], head_std=g.get("head_std", 0.05))
    target = LlamaForCausalLM.from_state_dict(LlamaConfig.from_dict(g["tcfg"]), tsd, device)
    draft = LlamaForCausalLM_68M.from_state_dict(LlamaConfig.from_dict(g["dcfg"]), dsd, device)
    gamma = g["gamma"]
    cache = FlashSimpleCache(target, g["prefill"] + g["gen_len"] + 16)
    gcache = RetrievalCache(target, max_budget=g["budget"], prefill=g["prefill"], gamma=gamma, chunk_size=g["chunk"])
    dcache = StreamingLLMEvictionCache(draft, start_size=16, recent_size=256 - 16 - gamma, gamma=gamma)
    ge = GraphInferenceEngine(target, cache, gcache, draft, dcache)
    T = g["temperature"] if temperature is None else temperature
    P = g["top_p"] if top_p is None else top_p
    if graphs:
        ge.initialize_cuda_graph(gamma, probs=True, temperature=T, top_p=P, verbose=False, capture_target=target_graph)
    else:
        ge.initialize_eager(gamma, probs=True, temperature=T, top_p=P)
    return ge


def prompt_of(g):
    return specs.random_prompt(g["tcfg"]["vocab_size"], g["prefill"], g["pseed"])


def fixed_uniforms(n=4096, seed=99):
    gen = torch.Generator().manual_seed(seed)
    return torch.rand(n, generator=gen).tolist()


def teacher_forced_gaps(g, stream, tsd=None, dsd=None):
    """Feed `stream` (first token + generated tokens) through the CPU oracle's greedy autoregressive path and
    return, for every token, how far its oracle logit is below the oracle's best logit given the same
    prefix (0 = it IS the oracle's argmax).  This is the device-independent statement of 'token-for-token
    at temp=0': rounding may only ever pick a token whose oracle logit is within fp16 noise of the best."""
    eng, _, _ = build_oracle(g)
    prompt = prompt_of(g)
    eng.kv_cache.reset()
    logits = eng.inference(prompt)[0, -1]
    gaps = [float(logits.max() - logits[stream[0]])]
    for i in range(len(stream) - 1):
        logits = eng.model.forward(torch.tensor([[stream[i]]]), eng.kv_cache, None)[0, -1]
        gaps.append(float(logits.max() - logits[stream[i + 1]]))
    return gaps


def common_prefix(a, b):
    n = 0
    for x, y in zip(a, b):
        if x != y:
            break
        n += 1
    return n


def force_reference_selection(ops_module, ref_idx_layers, log):
    """Teacher-force a retrieval build: wraps ``ops_module.retrieval_topk`` so that every call (one per layer, in order)
    records (own scores, own indices) in ``log`` and RETURNS the reference's indices of that layer (tests/golden/tp_world8.pt
    ``topk_idx[rank]``).  A near-tie at the k-th chunk score may resolve differently on scores that differ by fp16 noise, and
    one swapped chunk moves the retrieval-verify logits by more than any rounding tolerance; ``check_selection`` then judges
    the product's OWN choice tie-tolerantly.  Returns the undo function."""
    real = ops_module.retrieval_topk
    state = {"layer": 0}

    def wrapped(scores, sets):
        own = real(scores, sets)
        ref = ref_idx_layers[state["layer"]].to(device=own.device, dtype=own.dtype).reshape(own.shape)
        log.append((scores.detach().float().cpu(), own.detach().cpu().long(), ref.detach().cpu().long()))
        state["layer"] += 1
        return ref.contiguous()
    ops_module.retrieval_topk = wrapped
    return lambda: setattr(ops_module, "retrieval_topk", real)


def check_selection(log, spacings=3.0):
    """Every chunk the product selected but the reference did not (and vice versa) must sit within ``spacings`` fp16 spacings
    of the product's own k-th best score: the two selections differ only where the scores tie to rounding noise.  Returns
    the number of swapped chunks."""
    swapped = 0
    for layer, (scores, own, ref) in enumerate(log):
        for h in range(own.shape[0]):
            a, b = set(own[h].tolist()), set(ref[h].tolist())
            if a == b:
                continue
            kth = scores[h][own[h][1:]].min()                        # the product's k-th best score (chunk 0 is forced in)
            tol = spacings * max(abs(float(kth)), 6.1e-5) * 2.0 ** -10 + 1e-3
            for c in a ^ b:
                swapped += 1
                gap = abs(float(scores[h][c]) - float(kth))
                assert gap <= tol, f"layer {layer} head {h}: chunk {c} is {gap:.3e} from the k-th score {float(kth):.4f} (not a tie)"
    return swapped
