"""Op-level CPU restatement of the TriForce hot path (torch CPU, fp16 storage / fp32 math
exactly where the reference has it).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg as the *checker*; the product (triforce_amd/) never imports it.

Parity status: the reference ships no golden vectors for this path — it is
pinned here against the reference's own Python run in the build container
(oracle/gen_golden.py -> tests/golden/*.pt, checked by tests/test_oracle_golden.py).
Third-party arithmetic not in /root/reference: flash-attn 2.5.7 ``flash_attn_with_kvcache``
(README.md:115) — restated from its published semantics as fp32 softmax attention with a
bottom-right-aligned causal mask (``attn_kvcache`` below).

Every function cites the reference lines it follows (paths relative to /root/reference).
Logical layouts: hidden (rows, hidden); q/k/v (rows, H, D); KV caches (T, H, D).
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------
# RoPE tables
# ----------------------------------------------------------------------------------------
def rope_tables_plain(dim, max_pos, base=10000.0, dtype=torch.float16):
    """models/modeling_llama.py:21-41 (LlamaRotaryEmbedding): fp32 tables rounded to fp16."""
    inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
    t = torch.arange(max_pos, dtype=inv_freq.dtype)
    freqs = torch.outer(t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def _yarn_correction_dim(num_rot, dim, base, max_pos):
    # models/modeling_llama.py:55-56
    return (dim * math.log(max_pos / (num_rot * 2 * math.pi))) / (2 * math.log(base))


def rope_tables_yarn(dim, max_pos, factor, orig_max_pos, base=10000.0, beta_fast=32, beta_slow=1,
                     dtype=torch.float16):
    """models/modeling_llama.py:50-124 (LlamaYaRNRotaryEmbedding); base is hard-wired 10000
    at the call site (modeling_llama.py:193)."""
    pos_freqs = base ** (torch.arange(0, dim, 2).float() / dim)
    inv_extra = 1.0 / pos_freqs
    inv_inter = 1.0 / (factor * pos_freqs)
    low = max(math.floor(_yarn_correction_dim(beta_fast, dim, base, orig_max_pos)), 0)
    high = min(math.ceil(_yarn_correction_dim(beta_slow, dim, base, orig_max_pos)), dim - 1)
    lo, hi = float(low), float(high)
    if lo == hi:
        hi += 0.001
    ramp = torch.clamp((torch.arange(dim // 2, dtype=torch.float32) - lo) / (hi - lo), 0, 1)
    mask = (1 - ramp)
    inv_freq = inv_inter * (1 - mask) + inv_extra * mask
    mscale = 1.0 if factor <= 1 else 0.1 * math.log(factor) + 1.0
    t = torch.arange(max_pos, dtype=torch.float32)
    freqs = torch.outer(t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return (emb.cos() * mscale).to(dtype), (emb.sin() * mscale).to(dtype)


def head_dim_of(cfg):
    """hidden_size // num_attention_heads (modeling_llama.py:169), unless the config names ``head_dim`` explicitly: a
    tensor-parallel SHARD keeps the model's head_dim with fewer heads (TP_layers.py:115-116 computes head_dim from the
    UNSHARDED head count; oracle.specs.shard_of builds such configs)."""
    return int(cfg.get("head_dim") or cfg["hidden_size"] // cfg["num_attention_heads"])


def rope_tables_for(cfg):
    """Dispatch of modeling_llama.py:181-198 (_init_rope)."""
    D = head_dim_of(cfg)
    rs = cfg.get("rope_scaling")
    if rs is None:
        return rope_tables_plain(D, cfg["max_position_embeddings"], cfg["rope_theta"])
    if rs["type"] != "yarn":
        raise ValueError(f"Unknown RoPE scaling type {rs['type']}")
    return rope_tables_yarn(D, cfg["max_position_embeddings"], rs["factor"],
                            rs["original_max_position_embeddings"])


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rope(x, cos, sin, positions):
    """models/tensor_op.py:25-50 / modeling_llama_68m.py:30-38.  x (rows,H,D) fp16,
    positions (rows,) long.  All arithmetic in fp16: two rounded products, one rounded add."""
    c = cos[positions].unsqueeze(1)
    s = sin[positions].unsqueeze(1)
    return (x * c) + (rotate_half(x) * s)


# ----------------------------------------------------------------------------------------
# dense blocks
# ----------------------------------------------------------------------------------------
def rms_norm(x, weight, eps):
    """models/modeling_llama.py:138-143 == tensor_op.py:52-64: fp32 normalise, cast to fp16,
    THEN multiply by the fp16 weight."""
    dt = x.dtype
    xf = x.to(torch.float32)
    var = xf.pow(2).mean(-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    return weight * xf.to(dt)


def silu_mul(gate, up):
    """models/modeling_llama.py:156-159: act_fn(gate_proj(x)) * up_proj(x), fp16 in/out."""
    return F.silu(gate) * up


def linear(x, w):
    return F.linear(x, w)


SOFTMAX_SCALE_D128 = 0.08837890625  # fp16(1)/sqrt(fp16(128)) — modeling_llama.py:240
SOFTMAX_SCALE_D64 = 0.125


def softmax_scale_for(head_dim):
    """``1/torch.sqrt(torch.tensor(head_dim, dtype=torch.float16))`` (modeling_llama.py:240)."""
    return float(1 / torch.sqrt(torch.tensor(head_dim, dtype=torch.float16)))


def attn_kvcache(q, k, v, scale, causal=True):
    """flash_attn_with_kvcache(q, k_cache, v_cache, softmax_scale, causal=True), bsz 1, MHA.
    q (sq,H,D), k/v (sk,H,D) fp16 -> (sq,H,D) fp16.  fp32 scores/softmax/PV; bottom-right
    causal alignment: query i sees keys [0, sk-sq+i]."""
    sq, H, D = q.shape
    sk = k.shape[0]
    qf = q.float().permute(1, 0, 2)
    kf = k.float().permute(1, 2, 0)
    vf = v.float().permute(1, 0, 2)
    s = torch.matmul(qf, kf) * float(scale)
    if causal:
        qi = torch.arange(sq).view(sq, 1)
        kj = torch.arange(sk).view(1, sk)
        s = s.masked_fill(kj > (sk - sq + qi), float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = torch.matmul(p, vf)
    return o.permute(1, 0, 2).to(q.dtype).contiguous()


# ----------------------------------------------------------------------------------------
# retrieval cache build  (models/cache.py:146-178 == :517-556)
# ----------------------------------------------------------------------------------------
def retrieval_chunk_means(k, prefill, chunk):
    """cache.py:154: key_cache[:prefill].view(C, chunk, H, D).mean(dim=-3) -> (C,H,D) fp16."""
    T, H, D = k.shape
    C = prefill // chunk
    return k[:prefill].view(C, chunk, H, D).mean(dim=-3)


def retrieval_scores(k, q, prefill, chunk):
    """cache.py:154-157: un-scaled q . mean(K_chunk) in fp16 -> (H, C) fp16.  q (H,D) post-RoPE."""
    kbar = retrieval_chunk_means(k, prefill, chunk)                # (C,H,D)
    return torch.matmul(q.unsqueeze(1), kbar.permute(1, 2, 0)).squeeze(1)   # (H,1,D)x(H,D,C)


def _sortable_u16(x_fp16):
    """Total-order key for fp16 bit patterns: bigger float -> bigger key; +NaN above +inf."""
    b = x_fp16.view(torch.int16).to(torch.int32) & 0xFFFF
    neg = (b & 0x8000) != 0
    return torch.where(neg, (~b) & 0xFFFF, b | 0x8000)


def retrieval_topk(scores, select_sets):
    """cache.py:159-162: topk(scores[:,1:], k=select_sets-1) sorted descending, +1, chunk 0
    prepended.  torch.topk's order among equal scores is implementation-defined (SURVEY §7
    'Tie-breaking'); the canonical rule restated here and implemented by the HIP kernel is
    *descending score, ascending chunk index among equals*.  Returns (H, select_sets) int64."""
    H, C = scores.shape
    k = select_sets - 1
    key = _sortable_u16(scores[:, 1:]).to(torch.int64) * 65536 + (65535 - torch.arange(1, C).view(1, -1))
    order = torch.argsort(key, dim=-1, descending=True)[:, :k] + 1
    return torch.cat([torch.zeros(H, 1, dtype=torch.int64), order], dim=-1)


def retrieval_gather(kv, idx, chunk):
    """cache.py:163-175: per-head gather of whole chunks.  kv (T,H,D), idx (H,sets) ->
    (sets*chunk, H, D): slot j of head h = chunk idx[h,j]."""
    T, H, D = kv.shape
    sets = idx.shape[1]
    out = torch.empty(sets * chunk, H, D, dtype=kv.dtype)
    tok = (idx.unsqueeze(-1) * chunk + torch.arange(chunk).view(1, 1, chunk)).reshape(H, sets * chunk)
    for h in range(H):
        out[:, h] = kv[tok[h], h]
    return out


def topk_matches_reference(scores, ours, theirs):
    """Tie-tolerant comparison used to pin retrieval_topk against torch.topk (reference):
    identical score sequence, identical index set strictly above the k-th score."""
    H = scores.shape[0]
    for h in range(H):
        so, st = scores[h, ours[h, 1:]], scores[h, theirs[h, 1:]]
        if not torch.equal(so, st):
            return False
        thr = so[-1]
        a = set(ours[h, 1:][so > thr].tolist())
        b = set(theirs[h, 1:][st > thr].tolist())
        if a != b or ours[h, 0] != 0 or theirs[h, 0] != 0:
            return False
    return True


# ----------------------------------------------------------------------------------------
# sampling  (utils/sampling.py)
# ----------------------------------------------------------------------------------------
def norm_logits(logits, temperature=0.6, top_k=-1, top_p=0.9):
    """utils/sampling.py:43-60 with top_k_top_p_filter :5-27.  logits (rows,V) fp32."""
    assert logits.dim() == 2
    logits = logits / temperature
    if top_k > 0:
        kth = torch.topk(logits, min(top_k, logits.size(-1)))[0]
        logits = logits.masked_fill(logits < kth[:, [-1]], float("-inf"))
    if top_p > 0.0:
        sorted_logits, sorted_idx = torch.sort(logits, descending=True, stable=True)
        cum = torch.cumsum(F.softmax(sorted_logits, dim=-1), dim=-1)
        remove = cum > top_p
        remove[..., 1:] = remove[..., :-1].clone()
        remove[..., 0] = False
        remove = remove.scatter(1, sorted_idx, remove)
        logits = logits.masked_fill(remove, float("-inf"))
    return F.softmax(logits, dim=-1)


def max_fn(x):
    """utils/sampling.py:68-75: relu(x)/sum(relu(x))."""
    xm = torch.where(x > 0, x, torch.zeros_like(x))
    return xm / torch.sum(xm, dim=-1, keepdim=True)


def sample_inverse_cdf(probs, u):
    """Injected-RNG stand-in for torch.multinomial(probs, 1) (sampling.py:63-66): first index
    whose inclusive fp32 cumulative sum exceeds u * total.  Same distribution, explicit u."""
    c = torch.cumsum(probs.float(), dim=-1)
    tgt = float(u) * float(c[-1])
    idx = int(torch.searchsorted(c, torch.tensor(tgt, dtype=c.dtype), right=True))
    nz = torch.nonzero(probs > 0).flatten()
    if idx >= probs.numel() or probs[idx] <= 0:      # guard rounding at the tail / zero bins
        cand = nz[nz >= idx]
        idx = int(cand[0]) if cand.numel() else int(nz[-1])
    return idx


def accept_chain(target_probs, spec_probs, tokens, uniforms, inclusive=False):
    """The sequential accept loop of utils/decoding.py:97-121 (on-chip, ``r <``) /
    :342-371 (TP, ``r <=``): returns (count, flags) where flags[i] says token i passed
    ``r_i < min(1, p_i[x_i]/q_i[x_i])`` and count = length of the accepted prefix.
    target_probs (g2+1,V), spec_probs (g2,V), tokens (g2,), uniforms (g2,)."""
    g2 = len(tokens)
    flags = []
    for i in range(g2):
        t = int(tokens[i])
        ratio = target_probs[i, t] / spec_probs[i, t]
        m = torch.min(torch.tensor([1.0]), ratio.reshape(1))
        r = torch.as_tensor(uniforms[i], dtype=torch.float32).reshape(1)
        flags.append(bool(r <= m) if inclusive else bool(r < m))
    count = 0
    for f in flags:
        if not f:
            break
        count += 1
    return count, flags


def accept_and_correct(target_probs, spec_probs, tokens, uniforms, inclusive=False, eos_token_id=-1):
    """Whole outer-step decision of utils/decoding.py:96-134 as one function (the checker for
    tf_accept_chain): accept chain, stop right after an accepted eos unless it is the last drafted
    token, residual resample on rejection (max_fn(p-q), :114), bonus sample when all passed (:130).
    The sample uses uniforms[examined].  Returns (count, next_token, reason, consumed) with
    reason 0 = rejected, 1 = all accepted, 2 = eos accepted."""
    g2 = len(tokens)
    count, flags = accept_chain(target_probs, spec_probs, tokens, uniforms, inclusive)
    reason = 1 if count == g2 else 0
    for i in range(count):
        if int(tokens[i]) == eos_token_id and i + 1 < g2:
            count, reason = i + 1, 2
            break
    examined = count + 1 if reason == 0 else count
    if reason == 2:
        return count, int(eos_token_id), 2, examined
    if reason == 0:
        nxt = sample_inverse_cdf(max_fn(target_probs[count] - spec_probs[count]), uniforms[examined])
    else:
        nxt = sample_inverse_cdf(target_probs[g2], uniforms[examined])
    return count, nxt, reason, examined + 1


def middle_accept(p, q_d, d, n, uniforms):
    """One inner step of utils/decoding.py:190-220: accept test of drafted token d against p[n], then
    the follow-up sample from p[n+acc].  Returns (accepted, follow_up_token)."""
    ratio = p[n, d] / q_d[d]
    m = torch.min(torch.tensor([1.0]), ratio.reshape(1))
    acc = bool(torch.as_tensor(uniforms[0], dtype=torch.float32).reshape(1) < m)
    b = sample_inverse_cdf(p[n + (1 if acc else 0)], uniforms[1])
    return int(acc), b
