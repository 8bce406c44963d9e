"""Aligned synthetic weights: a (target, draft) pair whose next-token distributions agree to a tunable degree.

Why: no checkpoints exist offline.  Random-init weights make the three levels of the TriForce hierarchy
(68M draft -> target over the retrieval cache -> target over the full cache; reference utils/decoding.py:41-223)
disagree on almost every token (measured acceptance 0.008), which pins the decode loop at its worst case — gamma
inner iterations for ~1 token per outer step — so neither tokens/s nor "avg accepted length" describes the regime
the algorithm exists for (the reference's README.md:49-55 quotes 2.2x over autoregressive WITH trained models).
These weights keep every tensor at its real shape and dense (each kernel streams exactly the bytes it streams for a
trained checkpoint) and plant just enough structure to set the two acceptance rates:

  * hidden = S1 | S2 | S3 (h/2, h/4, h/4 coordinates).  Embeddings live in S1 and are the only block q/k/v and
    gate/up read; attention outputs (o_proj rows) are written to S2, MLP outputs (down_proj rows) to S3; everything
    else is dense noise a factor ``leak`` smaller.
  * lm_head columns over S1 carry a planted bigram table: token t has K successors succ_k(t) with logits
    peak + off_k(t) (one rank-1 term beta * unit(E[t]) per pair, then a few Jacobi sweeps remove the cross-talk between
    terms, so the logits at the planted pairs are exact for the fp16 weights).  Target and draft share the table for a
    fraction of the vocabulary and disagree (another random table) on the rest: that fraction sets the
    draft -> retrieval-model acceptance.
  * lm_head columns over S2 are gamma_r * U (U random unit rows): they read the summed attention outputs.  The q
    projection is scaled down (near-uniform attention), so what S2 holds is essentially the mean value vector of the
    keys a forward attended — small over the full 125K-token cache, larger and different over the 4K retrieval cache.
    gamma_r therefore turns "which keys were attended" into a logit perturbation of controlled size; it is calibrated
    after prefill (``calibrate``) by bisection so that the mean sum_v min(p_full, p_retrieval) over probe rows equals the
    requested retrieval -> target acceptance.  S3 is not read: the MLP runs at full cost and leaves the logits alone.

Spec string (``--weights``): ``aligned[:<draft_acc>[:<retrieval_acc>[:<seed>]]]`` (defaults 0.7 : 0.9 : 0).
"""
import dataclasses
import math
import zlib

import torch

from .. import ops


@dataclasses.dataclass
class AlignedSpec:
    draft_acc: float = 0.7        # requested acceptance of 68M-draft tokens by the retrieval-cache model
    retrieval_acc: float = 0.9    # requested acceptance of retrieval-model tokens by the full-cache model
    seed: int = 0
    candidates: int = 3           # planted successors per token
    peak: float = 10.0            # planted logit of the first successor (background logits stay ~N(0, <0.5))
    q_gain: float = 0.1           # scale of the q projection: near-uniform attention weights
    rel_attn: float = 0.1         # |sum of attention outputs| / |embedding| aimed at for a retrieval forward
    rel_mlp: float = 0.1          # same for the MLP outputs
    leak: float = 1.0 / 64        # scale of the "unused" blocks (dense noise instead of zeros)
    embed_std: float = 2.0

    def label(self):
        return f"aligned:{self.draft_acc}:{self.retrieval_acc}:{self.seed}"


def parse_spec(s):
    """``aligned[:draft_acc[:retrieval_acc[:seed]]]`` -> AlignedSpec, anything else -> None."""
    if not isinstance(s, str) or not (s == "aligned" or s.startswith("aligned:")):
        return None                     # exactly the spec grammar: a checkpoint directory named aligned* is a path
    parts = s.split(":")[1:]
    spec = AlignedSpec()
    if len(parts) > 0 and parts[0]:
        spec.draft_acc = float(parts[0])
    if len(parts) > 1 and parts[1]:
        spec.retrieval_acc = float(parts[1])
    if len(parts) > 2 and parts[2]:
        spec.seed = int(parts[2])
    if not (0.0 <= spec.draft_acc <= 1.0 and 0.0 < spec.retrieval_acc <= 1.0):
        raise ValueError(f"acceptance rates must lie in [0, 1]: {s!r}")
    return spec


def subspaces(hidden):
    n1, n2 = hidden // 2, hidden // 4
    return slice(0, n1), slice(n1, n1 + n2), slice(n1 + n2, hidden)


def plant(vocab, spec):
    """The planted successor tables (CPU tensors, a function of (vocab, spec) only):
    succ_t / succ_d (K, V) int64 successors for the target / the draft, off (K, V) logit offsets, agree (V,) bool."""
    g = torch.Generator().manual_seed(1000003 * spec.seed + 12345)
    K, M = spec.candidates, vocab - 3
    assert M > 4 * K, "vocabulary too small for the planted table"
    perm_t, perm_d = torch.randperm(M, generator=g), torch.randperm(M, generator=g)
    shifts = [0]
    while len(shifts) < K:                                   # distinct cyclic shifts -> distinct successors per token
        s = int(torch.randint(1, M, (1,), generator=g))
        if s not in shifts:
            shifts.append(s)
    idx = torch.arange(vocab).clamp(min=3) - 3               # special tokens 0..2 follow token 3's table
    succ_t = torch.stack([3 + perm_t[(idx + s) % M] for s in shifts])
    succ_alt = torch.stack([3 + perm_d[(idx + s) % M] for s in shifts])
    off = torch.zeros(K, vocab)
    for k in range(1, K):
        lo, width = (0.3, 2.2) if k == 1 else (0.5, 2.0)
        off[k] = off[k - 1] - (lo + width * torch.rand(vocab, generator=g))
    # tokens of the draft that follow the target's table.  On those the draft matches the FULL-cache target exactly; the
    # retrieval model it is tested against is the calibrated perturbation of that, so it accepts them at about
    # retrieval_acc: the planted fraction is raised accordingly.
    frac = min(1.0, spec.draft_acc / max(spec.retrieval_acc, 1e-6))
    agree = torch.rand(vocab, generator=g) < frac
    succ_d = torch.where(agree.unsqueeze(0), succ_t, succ_alt)
    return dict(succ_t=succ_t, succ_d=succ_d, off=off, agree=agree, frac=frac)


def _silu_mul_std(s):
    """std of silu(g) * u for independent g, u ~ N(0, s^2) (Monte Carlo, fixed seed)."""
    g = torch.Generator().manual_seed(7)
    a = torch.randn(200000, generator=g, dtype=torch.float64) * s
    b = torch.randn(200000, generator=g, dtype=torch.float64) * s
    return float(((a * torch.sigmoid(a)) * b).pow(2).mean().sqrt())


def _unit_rows(x):
    return x / x.norm(dim=1, keepdim=True).clamp_min(1e-12)


def _rms_rows(x16, eps):
    """RMSNorm with unit weight as the model computes it: fp32 normalise -> fp16 (modeling_llama.py:138-143)."""
    xf = x16.float()
    return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).to(torch.float16)


def fit_planted(W32, E16, succ, off, peak, eps, sweeps=4, chunk=2048):
    """Make  fp16(W) . rmsnorm(E[t])  equal  peak + off[k, t]  at every planted pair (t, succ[k, t]).
    W32 (V, h) fp32 master, updated in place; returns the largest remaining error of the last sweep.  Adding c * unit(E[t])
    to row v moves logit (t, v) by c * sqrt(h); the other pairs sharing row v move by O(c * sqrt(h / dim)) — a
    contraction, so a few Jacobi sweeps suffice."""
    V, h = W32.shape
    K = succ.shape[0]
    dev = W32.device
    succ, off = succ.to(dev), off.to(dev)
    worst = 0.0
    for sweep in range(sweeps + 1):
        W16 = W32.to(torch.float16).float()
        delta = torch.zeros_like(W32) if sweep < sweeps else None
        worst = 0.0
        for c0 in range(0, V, chunk):
            t = torch.arange(c0, min(V, c0 + chunk), device=dev)
            n = _rms_rows(E16[t], eps).float()
            L = (n @ W16.T).to(torch.float16).float()            # the model's fp16 GEMM output
            ehat = _unit_rows(E16[t].float())
            for k in range(K):
                res = (peak + off[k, t]) - L[torch.arange(t.numel(), device=dev), succ[k, t]]
                worst = max(worst, float(res.abs().max()))
                if delta is not None:
                    delta.index_add_(0, succ[k, t], (res / math.sqrt(h)).unsqueeze(1) * ehat)
        if delta is not None:
            W32 += delta
    return worst


def init_weights(W, spec, role, attn_keys=4096):
    """Fill a llama_core.LlamaWeights (any rank / world size) with the aligned construction.  role: "target" | "draft".
    attn_keys: the number of keys a retrieval-cache (target) / streaming-cache (draft) forward attends — only sets the
    scale of o_proj so that the attention outputs stay a ``rel_attn`` fraction of the embedding norm."""
    cfg, dev = W.cfg, W.device
    hid, V, L = cfg.hidden_size, cfg.vocab_size, W.L
    s1, s2, s3 = subspaces(hid)
    n1, n2, n3 = s1.stop - s1.start, s2.stop - s2.start, s3.stop - s3.start
    hd = W.H_local * W.D
    base_std = 0.02

    def draw(tag, *shape):
        g = torch.Generator(device=dev)
        g.manual_seed((spec.seed * 1000003 + zlib.crc32(repr((role, tag)).encode())) % (2 ** 31))
        return torch.randn(*shape, generator=g, device=dev, dtype=torch.float32)

    def row_scaled(x, sl, main, rest):
        scale = torch.full((x.shape[0], 1), rest, device=dev)
        scale[sl] = main
        return x * scale

    def col_scaled(x, sl, main, rest):
        scale = torch.full((1, x.shape[1]), rest, device=dev)
        scale[:, sl] = main
        return x * scale

    # scales: |E| ~ embed_std * sqrt(n1); attention / MLP outputs sized relative to it
    e_norm = spec.embed_std * math.sqrt(n1)
    v_std = base_std * math.sqrt(hid)                         # q/k/v entries for a unit-rms input
    a_std = v_std / math.sqrt(max(attn_keys, 1))              # near-uniform attention: mean of attn_keys value rows
    full_hd = W.H * W.D
    o_std = spec.rel_attn * e_norm / (math.sqrt(L) * math.sqrt(n2) * math.sqrt(full_hd) * a_std)
    act_std = _silu_mul_std(v_std)
    d_std = spec.rel_mlp * e_norm / (math.sqrt(L) * math.sqrt(n3) * math.sqrt(cfg.intermediate_size) * act_std)

    E = col_scaled(draw("embed", V, hid), s1, spec.embed_std, spec.embed_std * spec.leak).to(torch.float16)
    W.embed = E
    W.norm = torch.ones(hid, dtype=torch.float16, device=dev)
    for i in range(L):
        # q/k/v and gate/up READ the embedding block only: a layer's attention / MLP output is then a function of the
        # attended tokens' embeddings, not of what earlier layers wrote.  (Reading S2 closes a loop — the part of S2
        # common to all positions survives the attention average undiminished and is re-amplified by every layer: 32
        # layers of that drown the embedding.)
        qkv = col_scaled(draw(("qkv", i, W.rank), 3 * hd, hid), s1, base_std, base_std * spec.leak)
        qkv[:hd] *= spec.q_gain
        W.wqkv.append(qkv.to(torch.float16))
        W.wo.append(row_scaled(draw(("o", i, W.rank), hid, hd), s2, o_std, o_std * spec.leak).to(torch.float16))
        W.wgu.append(col_scaled(draw(("gu", i, W.rank), 2 * W.I_local, hid), s1, base_std,
                                base_std * spec.leak).to(torch.float16))
        W.wd.append(row_scaled(draw(("d", i, W.rank), hid, W.I_local), s3, d_std, d_std * spec.leak).to(torch.float16))
        W.ln1.append(torch.ones(hid, dtype=torch.float16, device=dev))
        W.ln2.append(torch.ones(hid, dtype=torch.float16, device=dev))

    # lm_head: planted table over S1 (+ exactness sweeps), attention read-out over S2 (target only), noise over S3
    tab = plant(V, spec)
    succ = tab["succ_t"] if role == "target" else tab["succ_d"]
    off = tab["off"]
    beta0 = spec.peak / math.sqrt(hid)
    head = draw("head_noise", V, hid) * (beta0 / math.sqrt(n1)) * spec.leak
    ehat = _unit_rows(E.float())
    for k in range(succ.shape[0]):
        coef = ((spec.peak + off[k]) / math.sqrt(hid)).to(dev).unsqueeze(1)
        head.index_add_(0, succ[k].to(dev), coef * ehat)
    U2 = None
    if role == "target":
        U2 = (draw("readout", V, n2) / math.sqrt(n2)).to(torch.float16)
        head[:, s2] = U2.float() * spec.leak                  # gamma_r = leak until calibrate() sets it
    err = fit_planted(head, E, succ, off, spec.peak, W.eps)
    W.lm_head = head.to(torch.float16)
    W.aligned = dict(spec=spec, role=role, U2=U2, readout_gain=spec.leak if role == "target" else 0.0,
                     planted_fit_err=err, o_std=o_std, d_std=d_std, plant_fraction=tab["frac"],
                     succ0=succ[0].clone())
    W.finalize()
    return W


# ---- calibration of the retrieval -> target acceptance -----------------------------------------------------------
def _acceptance(lf, lr, temperature, top_p):
    from ..utils.sampling import norm_logits
    pf = norm_logits(lf.contiguous(), temperature=temperature, top_k=-1, top_p=top_p)
    pr = norm_logits(lr.contiguous(), temperature=temperature, top_k=-1, top_p=top_p)
    return float(torch.minimum(pf, pr).sum(-1).mean())


@torch.inference_mode()
def calibrate(W, run_retrieval, run_full, q_len, temperature, top_p, n_probe=8, gain_max=64.0):
    """Set the read-out gain gamma_r of the target's lm_head so that the probe-row mean of sum_v min(p_full, p_retr)
    equals spec.retrieval_acc.  ``run_retrieval(ids)`` / ``run_full(ids)`` run the two forwards of the SAME q_len
    tokens at the current cache position (the caller rolls the full cache back); the final hidden states are read
    through ``W.capture``.  The lm_head is rewritten IN PLACE (captured hipGraphs keep pointing at it)."""
    info = W.aligned
    spec, U2 = info["spec"], info["U2"]
    assert info["role"] == "target" and U2 is not None
    dev = W.device
    hid = W.cfg.hidden_size
    _, s2, _ = subspaces(hid)
    succ0 = info["succ0"].to(dev)
    g = torch.Generator().manual_seed(spec.seed * 31 + 5)
    starts = torch.randint(3, W.cfg.vocab_size, (n_probe,), generator=g).tolist()
    xf, xr = [], []
    for t in starts:
        ids = [t]
        while len(ids) < q_len:
            ids.append(int(succ0[ids[-1]]))
        ids = torch.tensor([ids], dtype=torch.long, device=dev)
        W.capture = []
        run_retrieval(ids)
        xr.append(W.capture[-1])
        W.capture = []
        run_full(ids)
        xf.append(W.capture[-1])
    W.capture = None
    xf, xr = torch.cat(xf), torch.cat(xr)
    nf, nr = _rms_rows(xf, W.eps).float(), _rms_rows(xr, W.eps).float()
    head = W.lm_head.w
    keep = torch.ones(hid, dtype=torch.bool, device=dev)
    keep[s2] = False
    Wk = head[:, keep].float()
    Af, Ar = nf[:, keep] @ Wk.T, nr[:, keep] @ Wk.T
    U = U2.float()
    Bf, Br = nf[:, s2] @ U.T, nr[:, s2] @ U.T

    def acc(gain):
        lf = (Af + gain * Bf).to(torch.float16).float()
        lr = (Ar + gain * Br).to(torch.float16).float()
        return _acceptance(lf, lr, temperature, top_p)

    lo, hi = 0.0, gain_max
    a_hi = acc(hi)
    if a_hi > spec.retrieval_acc:                             # even the largest gain cannot push it that low
        gain = hi
    else:
        for _ in range(24):
            mid = 0.5 * (lo + hi)
            if acc(mid) > spec.retrieval_acc:
                lo = mid
            else:
                hi = mid
        gain = 0.5 * (lo + hi)
    head[:, s2] = (U * gain).to(torch.float16)
    W.lm_head.refresh_()
    info["readout_gain"] = gain
    rel = float((xr[:, s2].float() - xf[:, s2].float()).norm(dim=1).mean() / xf.float().norm(dim=1).mean())
    out = dict(readout_gain=round(gain, 4), probe_rows=int(xf.shape[0]), probe_acceptance=round(acc(gain), 4),
               probe_acceptance_gain0=round(acc(0.0), 4), s2_difference_over_norm=round(rel, 5),
               s2_full_over_norm=round(float(xf[:, s2].float().norm(dim=1).mean() / xf.float().norm(dim=1).mean()), 5),
               s3_over_norm=round(float(xf[:, subspaces(hid)[2]].float().norm(dim=1).mean()
                                        / xf.float().norm(dim=1).mean()), 5))
    info["calibration"] = out
    return out


def calibrate_engine(graph_engine, gamma, temperature, top_p, n_probe=8):
    """Calibration for the single-GPU engine (utils/graph_infer.GraphInferenceEngine) right after prefill."""
    eng = graph_engine.engine
    W = eng.model.weights
    S = eng.kv_cache.seq_len

    def run_retrieval(ids):
        pos = torch.arange(S, S + ids.shape[1], device=ids.device).unsqueeze(0)
        eng.model(input_ids=ids, kv_cache=eng.kv_cache, graph_cache=eng.graph_cache, position_ids=pos, spec=True)

    def run_full(ids):
        eng.model(input_ids=ids, kv_cache=eng.kv_cache, graph_cache=None)
        eng.kv_cache.seq_len = S                               # roll the probe back

    return calibrate(W, run_retrieval, run_full, gamma + 1, temperature, top_p, n_probe=n_probe)


def calibrate_llm(llm, gamma, temperature, top_p, n_probe=8):
    """Calibration for the tensor-parallel engine (models/TP_llama.DistributedLlama): identical on every rank."""
    W = llm.weights
    S = llm.kv_cache.seq_len

    def run_retrieval(ids):
        pos = torch.arange(S, S + ids.shape[1], device=ids.device).unsqueeze(0)
        llm.retrieval_inference(ids, pos)

    def run_full(ids):
        caps, llm._target_caps = getattr(llm, "_target_caps", {}), {}   # eager forward: the hook is not in the graphs
        try:
            llm.inference(input_ids=ids)
        finally:
            llm._target_caps = caps
        llm.kv_cache.seq_len = S

    return calibrate(W, run_retrieval, run_full, gamma + 1, temperature, top_p, n_probe=n_probe)
