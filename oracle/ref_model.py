"""Whole-path CPU restatement: models, the three KV caches, the engine surface and the
Autoregressive / TriForce / Middle_Spec decode loops (single-device, "on-chip" path).

TEST INFRASTRUCTURE ONLY (see oracle/ref_ops.py header for the rules and the parity status).
Pinned against the reference's own Python by oracle/gen_golden.py: logits bit-identical and
token streams identical on the seeded tiny configs of tests/golden/.

Layouts here are the reference's: caches are (L, T, H, D) fp16 (the reference's
[L,1,T,H,D] with the bsz==1 axis dropped).  Citations are relative to /root/reference.
"""
import math

import torch
import torch.nn.functional as F

from . import ref_ops as R


# ----------------------------------------------------------------------------------------
# caches
# ----------------------------------------------------------------------------------------
class FullCache:
    """FlashSimpleCache — models/cache.py:20-61."""

    def __init__(self, cfg, max_budget):
        L, H = cfg["num_hidden_layers"], cfg["num_key_value_heads"]
        D = R.head_dim_of(cfg)
        self.layers, self.max_budget, self.seq_len = L, max_budget, 0
        self.key_cache = torch.zeros(L, max_budget, H, D, dtype=torch.float16)
        self.value_cache = torch.zeros(L, max_budget, H, D, dtype=torch.float16)

    def reset(self):
        self.seq_len = 0
        self.key_cache.zero_()
        self.value_cache.zero_()

    def update(self, k, v, layer):                       # cache.py:46-61
        n = k.shape[0]
        self.key_cache[layer, self.seq_len:self.seq_len + n] = k
        self.value_cache[layer, self.seq_len:self.seq_len + n] = v
        kk = self.key_cache[layer, :self.seq_len + n]
        vv = self.value_cache[layer, :self.seq_len + n]
        if layer == self.layers - 1:
            self.seq_len += n
        return kk, vv


class RetrievalCacheO:
    """RetrievalCache — models/cache.py:117-198."""

    def __init__(self, cfg, max_budget, prefill, chunk_size=8, gamma=6):
        assert prefill % chunk_size == 0 and max_budget % chunk_size == 0   # cache.py:126-127
        L, H = cfg["num_hidden_layers"], cfg["num_key_value_heads"]
        D = R.head_dim_of(cfg)
        self.layers, self.chunk_size, self.prefill, self.gamma = L, chunk_size, prefill, gamma
        self.chunks = prefill // chunk_size
        self.select_sets = max_budget // chunk_size
        self.max_budget = max_budget
        self.real_budget = max_budget + gamma + 1
        self.key_cache = torch.zeros(L, self.real_budget, H, D, dtype=torch.float16)
        self.value_cache = torch.zeros(L, self.real_budget, H, D, dtype=torch.float16)
        self.init_graph = False
        self.last_scores = [None] * L     # kept for stage-wise parity tests
        self.last_idx = [None] * L

    def reset(self):                                      # cache.py:196-198 (init_graph is NOT cleared)
        self.key_cache.zero_()
        self.value_cache.zero_()

    def init_graph_cache(self, kv_cache, q, layer):       # cache.py:146-178
        assert q.shape[0] == 1
        scores = R.retrieval_scores(kv_cache.key_cache[layer], q[0], self.prefill, self.chunk_size)
        idx = R.retrieval_topk(scores, self.select_sets)
        self.last_scores[layer], self.last_idx[layer] = scores, idx
        self.key_cache[layer, :self.max_budget] = R.retrieval_gather(
            kv_cache.key_cache[layer, :self.prefill], idx, self.chunk_size)
        self.value_cache[layer, :self.max_budget] = R.retrieval_gather(
            kv_cache.value_cache[layer, :self.prefill], idx, self.chunk_size)
        if layer == self.layers - 1:
            self.init_graph = True

    def update_graph_cache_retrieval(self, kv_cache, q, layer):   # cache.py:191-194
        self.init_graph_cache(kv_cache, q, layer)
        g = kv_cache.seq_len - self.prefill
        self.value_cache[layer, self.max_budget - g:self.max_budget] = \
            kv_cache.value_cache[layer, self.prefill:kv_cache.seq_len]
        self.key_cache[layer, self.max_budget - g:self.max_budget] = \
            kv_cache.key_cache[layer, self.prefill:kv_cache.seq_len]

    def update_graph_cache(self, kv_cache):               # cache.py:180-182
        g = kv_cache.seq_len - self.prefill
        self.value_cache[:, self.max_budget - g:self.max_budget] = \
            kv_cache.value_cache[:, self.prefill:kv_cache.seq_len].clone()
        self.key_cache[:, self.max_budget - g:self.max_budget] = \
            kv_cache.key_cache[:, self.prefill:kv_cache.seq_len].clone()

    def update(self, k, v, layer):                        # cache.py:184-189
        self.key_cache[layer, self.real_budget - self.gamma - 1:] = k
        self.value_cache[layer, self.real_budget - self.gamma - 1:] = v
        return self.key_cache[layer, :self.real_budget], self.value_cache[layer, :self.real_budget]


class StreamingCacheO:
    """StreamingLLMEvictionCache — models/cache.py:200-265.  Keys are stored UN-rotated."""

    def __init__(self, cfg, gamma=6, start_size=16, recent_size=496):
        L, H = cfg["num_hidden_layers"], cfg["num_key_value_heads"]
        D = R.head_dim_of(cfg)
        self.layers, self.gamma, self.start_size, self.recent_size = L, gamma, start_size, recent_size
        self.real_budget = start_size + recent_size + gamma + 3
        self.seq_len = 0
        self.key_cache = torch.zeros(L, self.real_budget, H, D, dtype=torch.float16)
        self.value_cache = torch.zeros(L, self.real_budget, H, D, dtype=torch.float16)

    def reset(self):                                      # cache.py:247-250 (seq_len is NOT reset)
        self.key_cache.zero_()
        self.value_cache.zero_()

    def update(self, k, v, layer):                        # cache.py:222-235
        n = k.shape[0]
        assert self.seq_len + n <= self.start_size + self.recent_size
        self.key_cache[layer, self.seq_len:self.seq_len + n] = k
        self.value_cache[layer, self.seq_len:self.seq_len + n] = v
        kk = self.key_cache[layer, :self.seq_len + n]
        vv = self.value_cache[layer, :self.seq_len + n]
        if layer == self.layers - 1:
            self.seq_len += n
        return kk, vv

    def spec_update(self, k, v, layer):                   # cache.py:237-245
        start = self.real_budget - self.gamma - 3
        end = start + k.shape[0]
        self.key_cache[layer, start:end] = k
        self.value_cache[layer, start:end] = v
        return self.key_cache[layer, :end], self.value_cache[layer, :end]

    def evict_prefill(self, incoming):                    # cache.py:252-261
        if self.seq_len + incoming <= self.start_size + self.recent_size:
            return
        keep = self.recent_size - incoming
        s = self.start_size
        self.key_cache[:, s:s + keep] = self.key_cache[:, self.seq_len - keep:self.seq_len].clone()
        self.value_cache[:, s:s + keep] = self.value_cache[:, self.seq_len - keep:self.seq_len].clone()
        self.seq_len = self.start_size + self.recent_size - incoming

    def evict_for_spec(self, cur):                        # cache.py:263-265
        s, r = self.start_size, self.recent_size
        self.key_cache[:, s:s + r] = self.key_cache[:, cur - r:cur].clone()
        self.value_cache[:, s:s + r] = self.value_cache[:, cur - r:cur].clone()


# ----------------------------------------------------------------------------------------
# models
# ----------------------------------------------------------------------------------------
class _Weights:
    def __init__(self, cfg, sd):
        self.cfg = cfg
        self.L = cfg["num_hidden_layers"]
        self.H = cfg["num_attention_heads"]
        self.D = R.head_dim_of(cfg)
        self.eps = cfg["rms_norm_eps"]
        self.sd = sd

    def layer(self, i, name):
        return self.sd[f"model.layers.{i}.{name}.weight"]


class OracleTarget(_Weights):
    """models/modeling_llama.py LlamaForCausalLM.forward (:384-414) and below."""

    def __init__(self, cfg, sd):
        super().__init__(cfg, sd)
        self.cos, self.sin = R.rope_tables_for(cfg)
        self.scale = R.softmax_scale_for(self.D)
        self.device = torch.device("cpu")

    def forward(self, input_ids, kv_cache, graph_cache=None, position_ids=None, spec=False):
        q_len = input_ids.shape[1]
        if position_ids is None:                                    # modeling_llama.py:345-349
            position_ids = torch.arange(kv_cache.seq_len, kv_cache.seq_len + q_len).unsqueeze(0)
        pos = position_ids[0]
        x = F.embedding(input_ids[0], self.sd["model.embed_tokens.weight"])   # (q, hidden)
        for i in range(self.L):
            res = x
            h = R.rms_norm(x, self.layer(i, "input_layernorm"), self.eps)
            q = R.linear(h, self.layer(i, "self_attn.q_proj")).view(q_len, self.H, self.D)
            k = R.linear(h, self.layer(i, "self_attn.k_proj")).view(q_len, self.H, self.D)
            v = R.linear(h, self.layer(i, "self_attn.v_proj")).view(q_len, self.H, self.D)
            q = R.apply_rope(q, self.cos, self.sin, pos)            # modeling_llama.py:221-224
            k = R.apply_rope(k, self.cos, self.sin, pos)
            if spec:                                                # :226-227
                kk, vv = graph_cache.update(k, v, i)
            else:                                                   # :228-238
                kk, vv = kv_cache.update(k, v, i)
                if q_len == 1 and isinstance(graph_cache, RetrievalCacheO):
                    if not graph_cache.init_graph:
                        graph_cache.init_graph_cache(kv_cache, q, i)
                    else:
                        graph_cache.update_graph_cache_retrieval(kv_cache, q, i)
            a = R.attn_kvcache(q, kk, vv, self.scale, causal=True)  # :240
            a = R.linear(a.reshape(q_len, self.H * self.D), self.layer(i, "self_attn.o_proj"))
            x = res + a
            res = x
            h = R.rms_norm(x, self.layer(i, "post_attention_layernorm"), self.eps)
            m = R.silu_mul(R.linear(h, self.layer(i, "mlp.gate_proj")), R.linear(h, self.layer(i, "mlp.up_proj")))
            x = res + R.linear(m, self.layer(i, "mlp.down_proj"))
        x = R.rms_norm(x, self.sd["model.norm.weight"], self.eps)
        return R.linear(x, self.sd["lm_head.weight"]).float().unsqueeze(0)   # (1,q,V) fp32


class OracleDraft(_Weights):
    """models/modeling_llama_68m.py LlamaForCausalLM.forward; attention :129-190."""

    def __init__(self, cfg, sd):
        super().__init__(cfg, sd)
        self.cos, self.sin = R.rope_tables_plain(self.D, cfg["max_position_embeddings"], cfg["rope_theta"])
        self.scale = R.softmax_scale_for(self.D)
        self.device = torch.device("cpu")

    def forward(self, input_ids, kv_cache, graph_cache=None, gamma_offset=-1):
        q_len = input_ids.shape[1]
        pos = torch.arange(kv_cache.seq_len, kv_cache.seq_len + q_len)       # 68m.py:286-291
        x = F.embedding(input_ids[0], self.sd["model.embed_tokens.weight"])
        for i in range(self.L):
            res = x
            h = R.rms_norm(x, self.layer(i, "input_layernorm"), self.eps)
            q = R.linear(h, self.layer(i, "self_attn.q_proj")).view(q_len, self.H, self.D)
            k = R.linear(h, self.layer(i, "self_attn.k_proj")).view(q_len, self.H, self.D)
            v = R.linear(h, self.layer(i, "self_attn.v_proj")).view(q_len, self.H, self.D)
            if gamma_offset >= 0:                                            # 68m.py:151-162
                kk, vv = graph_cache.spec_update(k, v, i)
                kv_len = gamma_offset + graph_cache.start_size + graph_cache.recent_size + 1
                qpos = torch.arange(graph_cache.real_budget - graph_cache.gamma - 3,
                                    graph_cache.real_budget - graph_cache.gamma + gamma_offset - 2)
                q = R.apply_rope(q, self.cos, self.sin, qpos)
                kk = R.apply_rope(kk, self.cos, self.sin, torch.arange(kv_len))
            else:                                                            # 68m.py:164-178
                kv_len = q_len + kv_cache.seq_len
                kk, vv = kv_cache.update(k, v, i)
                q = R.apply_rope(q, self.cos, self.sin, pos)
                kk = R.apply_rope(kk, self.cos, self.sin, torch.arange(kv_len))
            a = R.attn_kvcache(q, kk, vv, self.scale, causal=True)           # 68m.py:186
            a = R.linear(a.reshape(q_len, self.H * self.D), self.layer(i, "self_attn.o_proj"))
            x = res + a
            res = x
            h = R.rms_norm(x, self.layer(i, "post_attention_layernorm"), self.eps)
            m = R.silu_mul(R.linear(h, self.layer(i, "mlp.gate_proj")), R.linear(h, self.layer(i, "mlp.up_proj")))
            x = res + R.linear(m, self.layer(i, "mlp.down_proj"))
        x = R.rms_norm(x, self.sd["model.norm.weight"], self.eps)
        return R.linear(x, self.sd["lm_head.weight"]).float().unsqueeze(0)


# ----------------------------------------------------------------------------------------
# engine (utils/graph_infer.py) — eager; the graphs only replay these same calls
# ----------------------------------------------------------------------------------------
class OracleEngine:
    def __init__(self, target, kv_cache, graph_cache, draft, draft_cache, temperature=0.6, top_p=0.9,
                 prefill_chunk=128, draft_chunk=64, draft_temperature=None, draft_top_p=None):
        self.model, self.kv_cache, self.graph_cache = target, kv_cache, graph_cache
        self.draft, self.draft_cache = draft, draft_cache
        self.temperature, self.top_p = temperature, top_p
        self.prefill_chunk, self.draft_chunk = prefill_chunk, draft_chunk
        # TP path: DistributedLlama.draft_run is always called with its defaults temperature=0.6, top_p=0.9
        # (TP_llama.py:117, call sites decoding.py:316,394,453) and prefills the draft 128 tokens at a time (:119-126)
        self.draft_temperature = temperature if draft_temperature is None else draft_temperature
        self.draft_top_p = top_p if draft_top_p is None else draft_top_p

    def inference(self, input_ids):                       # graph_infer.py:29-41 (model_run)
        if input_ids.shape[-1] > 64:
            for i in range(math.ceil(input_ids.shape[1] / self.prefill_chunk)):
                logits = self.model.forward(input_ids[:, i * self.prefill_chunk:(i + 1) * self.prefill_chunk],
                                            self.kv_cache, None)
        else:
            logits = self.model.forward(input_ids, self.kv_cache, self.graph_cache)
        return logits

    def _draft_run(self, input_ids, gamma_offset=0, probs=False):   # graph_infer.py:43-58
        if input_ids.shape[-1] > 64:
            c = self.draft_chunk
            for i in range(math.ceil(input_ids.shape[1] / c)):
                self.draft_cache.evict_prefill(c)
                logits = self.draft.forward(input_ids[:, i * c:(i + 1) * c], self.draft_cache, None)
        else:
            logits = self.draft.forward(input_ids, self.draft_cache, self.draft_cache, gamma_offset=gamma_offset)
        if probs:
            return R.norm_logits(logits[0], temperature=self.draft_temperature, top_k=-1, top_p=self.draft_top_p)[-1]
        return logits

    def graph_draft_prefill(self, input_ids):
        return self._draft_run(input_ids)

    def graph_draft_inference(self, input_ids, gamma_offset=0):
        return self._draft_run(input_ids, gamma_offset=gamma_offset, probs=True)

    def graph_verify(self, input_ids, position_ids):      # graph_infer.py:61-67 (model_verify)
        logits = self.model.forward(input_ids, self.kv_cache, self.graph_cache, position_ids=position_ids, spec=True)
        return R.norm_logits(logits[0], temperature=self.temperature, top_k=-1, top_p=self.top_p)

    def update_graph_cache(self):
        self.graph_cache.update_graph_cache(self.kv_cache)


# ----------------------------------------------------------------------------------------
# RNG policies
# ----------------------------------------------------------------------------------------
class TorchRng:
    """Draws exactly like the reference (torch.multinomial / torch.rand(1) on the global CPU
    generator, lazily, in program order) so a seeded reference run is reproduced bit-for-bit."""

    def sample(self, probs):
        return int(torch.multinomial(probs.reshape(1, -1), num_samples=1, replacement=True))

    def uniform(self):
        return float(torch.rand(1))


class InjectedRng:
    """Explicit uniform stream shared with the GPU product path: every sample and every
    accept test consumes the next number; sampling is inverse-CDF (ref_ops.sample_inverse_cdf)."""

    def __init__(self, uniforms):
        self.u = [float(x) for x in uniforms]
        self.i = 0

    def _next(self):
        v = self.u[self.i % len(self.u)]
        self.i += 1
        return v

    def sample(self, probs):
        return R.sample_inverse_cdf(probs.reshape(-1), self._next())

    def uniform(self):
        return self._next()


# ----------------------------------------------------------------------------------------
# decode loops (utils/decoding.py)
# ----------------------------------------------------------------------------------------
def autoregressive(engine, input_ids, max_len, temperature, top_p, rng=None, top_k=-1, trace=None):
    """utils/decoding.py:14-37.  Returns the emitted token list (first token + max_len more)."""
    rng = rng or TorchRng()
    engine.kv_cache.reset()
    logits = engine.inference(input_ids)
    tok = rng.sample(R.norm_logits(logits[:, -1, :][0:1], temperature, top_k, top_p)[0])
    out = [tok]
    n = 0
    while n < max_len:
        logits = engine.model.forward(torch.tensor([[tok]]), engine.kv_cache, None)
        if trace is not None:
            trace.append(logits[0, -1].clone())
        tok = rng.sample(R.norm_logits(logits[:, -1, :][0:1], temperature, top_k, top_p)[0])
        out.append(tok)
        n += 1
    return out


def middle_spec(next_token, engine, gamma, rng):
    """utils/decoding.py:163-223 (68M drafts for the retrieval-cache model)."""
    n = 0
    ids = [int(next_token)]
    probs_rows = []
    accepted = drafted = 0
    verify_tokens = torch.full((1, gamma + 1), 100, dtype=torch.long)
    verify_tokens[0, 0] = int(next_token)
    S = engine.kv_cache.seq_len
    position_ids = torch.arange(S, S + gamma + 1).unsqueeze(0)
    while n < gamma:
        q_d = engine.graph_draft_inference(verify_tokens[:, :n + 1], gamma_offset=n)
        d = rng.sample(q_d)
        drafted += 1
        verify_tokens[0, n + 1] = d
        p = engine.graph_verify(verify_tokens, position_ids)
        r = rng.uniform()
        ratio = p[n, d] / q_d[d]
        if torch.tensor([r]) < torch.min(torch.tensor([1.0]), ratio.reshape(1)):   # :193
            probs_rows.append(p[n])
            ids.append(d)
            accepted += 1
            n += 1
            b = rng.sample(p[n])
            probs_rows.append(p[n])
            ids.append(b)
            n += 1
            if n <= gamma:      # decoding.py:209 writes the slice [n:n+1], empty (a no-op) when n == gamma+1
                verify_tokens[0, n] = b
        else:                                                                       # :211-220
            b = rng.sample(p[n])
            probs_rows.append(p[n])
            ids.append(b)
            n += 1
            verify_tokens[0, n] = b
    return ids, probs_rows, accepted / drafted


def triforce(engine, input_ids, gamma, max_len, temperature, top_p, rng=None, top_k=-1, eos_token_id=2,
             trace=None, dist=False):
    """utils/decoding.py:41-160.  Returns dict(tokens, acceptance_rate, accepted, drafted, n, counts).

    dist=True restates TriForce_Dist (:291-428) with Middle_Spec_Dist (:432-495) at world size 1; it differs from the
    on-chip loop in exactly three places: the outer accept test is ``r <=`` (:347, on-chip ``r <`` :99), the loop ENDS
    when the token that closed the accept scan is eos (:382-383 — before the cache updates and before a bonus token;
    the on-chip loop keeps generating), and the result is (accepted / drafted * gamma, seconds per token) (:426-428).
    The engine carries the other two (draft sampled at 0.6 / 0.9, draft prefill in 128-token blocks)."""
    rng = rng or TorchRng()
    engine.kv_cache.reset()
    engine.graph_cache.reset()
    engine.draft_cache.reset()
    engine.inference(input_ids[:, :-1])
    logits = engine.inference(input_ids[:, -1:])
    engine.graph_draft_prefill(input_ids)

    accepted_count = draft_count = 0
    next_token = rng.sample(R.norm_logits(logits[:, -1, :][0:1], temperature, top_k, top_p)[0])
    tokens = [next_token]
    counts = []
    n = 0
    while n < max_len:
        ids, spec_probs, _ = middle_spec(next_token, engine, gamma, rng)
        generated = ids[1:]
        draft_count += len(spec_probs)
        g2 = len(generated)
        verify_tokens = torch.tensor([[next_token] + generated], dtype=torch.long)
        logits = engine.inference(verify_tokens)                                    # :85
        probs = R.norm_logits(logits[0], temperature, top_k, top_p)
        if trace is not None:
            trace.append(dict(verify_tokens=verify_tokens.clone(), logits=logits[0].clone()))
        count = 0
        pass_tokens = torch.full((1, g2 + 2), 100, dtype=torch.long)
        pass_tokens[0, 0] = next_token
        pred = next_token
        for i, q_row, p_row in zip(generated, spec_probs, probs):                    # :96-121
            r = rng.uniform()
            bound = torch.min(torch.tensor([1.0]), (p_row[i] / q_row[i]).reshape(1))
            if (torch.tensor([r]) <= bound) if dist else (torch.tensor([r]) < bound):
                count += 1
                accepted_count += 1
                n += 1
                pred = i
                pass_tokens[0, count] = i
                tokens.append(i)
                if eos_token_id == i:
                    draft_count -= g2 - count
                    break
            else:
                n += 1
                pred = rng.sample(R.max_fn(p_row - q_row))
                pass_tokens[0, count + 1] = pred
                tokens.append(pred)
                break
            if eos_token_id == pred:
                break
        if dist and eos_token_id == pred:                                           # :382-383
            break
        engine.kv_cache.seq_len -= (g2 - count)                                     # :124
        engine.update_graph_cache()                                                 # :125
        if count == g2:                                                             # :127-134
            n += 1
            pred = rng.sample(probs[g2])
            pass_tokens[0, count + 1] = pred
            tokens.append(pred)
            count += 1
        counts.append(count)
        engine.graph_draft_inference(pass_tokens, gamma_offset=g2 + 1)              # :137
        dc = engine.draft_cache
        dc.evict_for_spec(dc.start_size + dc.recent_size + count)                   # :138-139
        next_token = pred
    return dict(tokens=tokens, acceptance_rate=accepted_count / draft_count, accepted=accepted_count,
                drafted=draft_count, n=n, counts=counts, avg_tokens=accepted_count / draft_count * gamma)
