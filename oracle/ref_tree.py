"""CPU restatement of the Sequoia tree path: utils/SpecTree_TP.py (SpecTree), models/TP_llama_tree.py
(retrieval_tree_inference, inference with attention_mask), models/tensor_op.py:121-181,230-272 (the SDPA
branches), models/cache.py:333-343 (gather_kv_incremental) and :385-483 (DistributedRetrievalCache_Seqouia),
test/offloading_seqouia.py:24-39 (residual, sampling without replacement).

TEST INFRASTRUCTURE ONLY (rules in oracle/ref_ops.py).  Single-process (TP = 1): the reference's all-reduce
over one rank is the identity.  Pinned by oracle/gen_golden.py `sequoia_case`, which runs the UNMODIFIED
reference SpecTree + TP_llama_tree.DistributedLlama + tensor_op on CPU (torch proxies for the CUDA-only calls)
on seeded tiny models and requires identical token streams and accept lists from this restatement.

Layouts are the reference's: caches (L, T, H, D) fp16 (bsz axis dropped).  Citations relative to /root/reference.
"""
import math

import torch
import torch.nn.functional as F

from . import ref_model as M
from . import ref_ops as R


# ----------------------------------------------------------------------------------------
# ops
# ----------------------------------------------------------------------------------------
def additive_tree_mask(mask01, dtype=torch.float16):
    """SpecTree_TP.py:65-67: 0 where visible, finfo(dtype).min where hidden."""
    m = (mask01 == 0).to(dtype)
    m.masked_fill_(m > 0, torch.finfo(dtype).min)
    return m


def attn_sdpa(q, k, v, attn_mask):
    """tensor_op.py:171,265: F.scaled_dot_product_attention(q, k, v, attn_mask=mask.half()) — default scale
    1/sqrt(D) (NOT the fp16-rounded flash-attn scale of the chain path).  q (n,H,D), k/v (T,H,D), mask (n,T)."""
    qq = q.permute(1, 0, 2).unsqueeze(0)
    kk = k.permute(1, 0, 2).unsqueeze(0)
    vv = v.permute(1, 0, 2).unsqueeze(0)
    o = F.scaled_dot_product_attention(qq, kk, vv, attn_mask=attn_mask.half()[None, None])
    return o[0].permute(1, 0, 2).contiguous()            # (n,H,D)


def residual(p, q):
    """offloading_seqouia.py:24-27."""
    r = (p - q).relu_()
    return r / (r.sum(dim=-1).unsqueeze(-1))


def sample_without_replacement(logits, rand, num_samples, temperature):
    """offloading_seqouia.py:29-39 (rank 0 branch)."""
    q = torch.softmax(logits / temperature, dim=-1)
    return (rand.log() / q).topk(k=num_samples).indices.flatten()


def accept_walk(target_probs, draft_logits, tokens, successors, temperature, rng):
    """SpecTree_TP.py:147-199 (accept_step + the walk of verify) on CPU.  rng.uniform() per examined child,
    rng.sample(residual) at the end.  Returns (accept_list, next_token or None, terminal, acc_count)."""
    draft_logits = draft_logits.clone()
    accept_list, acc_count, terminal = [0], 0, False
    res = None
    while True:
        node = accept_list[-1]
        p = target_probs[node]
        dl = draft_logits[node]
        children = successors[node]
        pos = -2
        if len(children) == 0:
            res = p
        else:
            pos = -1
            for c in children:
                tok = int(tokens[c])
                q = torch.softmax(dl / temperature, dim=-1)
                r = rng.uniform()
                if p[tok] > r * q[tok]:
                    pos = c
                    break
                p = residual(p.clone(), q)
                dl[tok] = torch.finfo(torch.float32).min
            if pos == -1:
                res = p
        if pos > -1:
            accept_list.append(pos)
            acc_count += 1
            if int(tokens[pos]) == 0 or int(tokens[pos]) == 2:
                terminal = True
                break
        else:
            break
    next_token = None
    if not terminal:
        if torch.isnan(res).any():
            terminal = True
        else:
            next_token = rng.sample(res)
            acc_count += 1
    return accept_list, next_token, terminal, acc_count


def grow_map_from_branches(branches):
    """The expansion loop of tree/tree_search.py:90-128 restated for a given list of per-level child counts:
    ids are handed out level by level, a node's mask row = its parent's row + itself."""
    roots, succ, depth, parents, n = [[0]], [[]], [0], [-1], 1
    out_branches = []
    level = [0]
    for blist in branches:
        blist = [int(b) for b in blist]
        assert len(blist) == len(level)
        out_branches.append(blist)
        nxt = []
        for node, b in zip(level, blist):
            kids = list(range(n, n + b))
            succ[node].extend(kids)
            succ.extend([[] for _ in range(b)])
            parents.extend([node] * b)
            depth.extend([depth[node] + 1] * b)
            nxt.extend(kids)
            n += b
        if not nxt:
            break
        roots.append(nxt)
        level = nxt
    if len(out_branches) < len(roots):
        out_branches.append([0] * len(roots[-1]))
    mask = torch.zeros(n, n, dtype=torch.long)
    for i in range(n):
        if parents[i] != -1:
            mask[i] = mask[parents[i]]
        mask[i][i] = 1
    return {"roots": roots, "branches": out_branches, "Successors": succ, "mask": mask,
            "depth": torch.LongTensor(depth), "size": n}


# ----------------------------------------------------------------------------------------
# caches
# ----------------------------------------------------------------------------------------
class FullCacheTree(M.FullCache):
    """DistributedSimpleCache with every layer on the device + gather_kv_incremental (cache.py:268-343).
    seq_len is advanced by the engine after the last layer (cache.py:349-351 / TP_llama_tree.py:207)."""

    def update(self, k, v, layer):                        # cache.py:310-318: no seq_len bump here
        n = k.shape[0]
        self.key_cache[layer, self.seq_len:self.seq_len + n] = k
        self.value_cache[layer, self.seq_len:self.seq_len + n] = v
        return self.key_cache[layer, :self.seq_len + n], self.value_cache[layer, :self.seq_len + n]

    def gather_kv_incremental(self, indices, offset):     # cache.py:333-343
        idx = [i + offset for i in indices]
        self.key_cache[:, offset:offset + len(idx)] = self.key_cache[:, idx].clone()
        self.value_cache[:, offset:offset + len(idx)] = self.value_cache[:, idx].clone()
        self.seq_len = offset + len(idx)


class RetrievalCacheSeq(M.RetrievalCacheO):
    """DistributedRetrievalCache_Seqouia — cache.py:385-483."""

    def __init__(self, cfg, max_budget, prefill, chunk_size=8, tree_size=128):
        super().__init__(cfg, max_budget, prefill, chunk_size, gamma=tree_size - 1)
        self.tree_size = tree_size
        assert self.real_budget == max_budget + tree_size

    def reset(self):                                      # :472-475 (init_graph IS cleared here)
        super().reset()
        self.init_graph = False

    def init_graph_cache(self, kv_cache, q, layer):       # :424-455
        if self.init_graph:
            raise ValueError("Graph is already initialized")
        super().init_graph_cache(kv_cache, q, layer)

    def update(self, k, v, layer, storage_ids):           # :458-466
        ids = torch.as_tensor(storage_ids, dtype=torch.long)
        self.key_cache[layer].index_copy_(0, ids, k)
        self.value_cache[layer].index_copy_(0, ids, v)
        return self.key_cache[layer], self.value_cache[layer]


# ----------------------------------------------------------------------------------------
# engine (models/TP_llama_tree.py, world_size 1)
# ----------------------------------------------------------------------------------------
class TreeEngine:
    def __init__(self, cfg, sd, prefill, gen_len, budget, chunk, tree_size):
        self.w = M.OracleTarget(cfg, sd)
        self.cfg = cfg
        self.device = torch.device("cpu")
        self.kv_cache = FullCacheTree(cfg, prefill + gen_len + tree_size)            # TP_llama_tree.py:70
        self.retrieval_cache = RetrievalCacheSeq(cfg, budget, prefill, chunk, tree_size)
        self.prefill_len = prefill

    def reset(self):
        self.kv_cache.reset()
        self.retrieval_cache.reset()

    def _forward(self, input_ids, position_ids, attend):
        """Shared layer loop (layer_compute :114-169 / layer_tree_speculation :292-346); `attend(i, q, k, v)`
        does the cache update + attention of layer i and returns (n, H*D)."""
        w = self.w
        n = input_ids.shape[1]
        pos = position_ids[0]
        x = F.embedding(input_ids[0], w.sd["model.embed_tokens.weight"])
        for i in range(w.L):
            res = x
            h = R.rms_norm(x, w.layer(i, "input_layernorm"), w.eps)
            q = R.linear(h, w.layer(i, "self_attn.q_proj")).view(n, w.H, w.D)
            k = R.linear(h, w.layer(i, "self_attn.k_proj")).view(n, w.H, w.D)
            v = R.linear(h, w.layer(i, "self_attn.v_proj")).view(n, w.H, w.D)
            q = R.apply_rope(q, w.cos, w.sin, pos)
            k = R.apply_rope(k, w.cos, w.sin, pos)
            a = attend(i, q, k, v)
            x = res + R.linear(a.reshape(n, w.H * w.D), w.layer(i, "self_attn.o_proj"))
            res = x
            h = R.rms_norm(x, w.layer(i, "post_attention_layernorm"), w.eps)
            m = R.silu_mul(R.linear(h, w.layer(i, "mlp.gate_proj")), R.linear(h, w.layer(i, "mlp.up_proj")))
            x = res + R.linear(m, w.layer(i, "mlp.down_proj"))
        x = R.rms_norm(x, w.sd["model.norm.weight"], w.eps)
        return R.linear(x, w.sd["lm_head.weight"]).float().unsqueeze(0)

    def inference(self, input_ids, position_ids=None, attention_mask=None, retrieval_cache=None):
        """TP_llama_tree.py:179-219 + tensor_op.TP_Attention :121-181.  attention_mask: dense (n, S+n) additive."""
        kvc = self.kv_cache
        n = input_ids.shape[1]
        if position_ids is None:
            position_ids = (kvc.seq_len + torch.arange(n)).unsqueeze(0)

        def attend(i, q, k, v):
            kk, vv = kvc.update(k, v, i)
            if retrieval_cache is not None:               # tensor_op.py:161-162
                retrieval_cache.init_graph_cache(kvc, q, i)
            if attention_mask is None:                    # :164-168 flash-attn, fp16-rounded scale
                return R.attn_kvcache(q, kk, vv, self.w.scale, causal=True)
            return attn_sdpa(q, kk, vv, attention_mask)   # :169-172
        logits = self._forward(input_ids, position_ids, attend)
        kvc.seq_len += n                                  # copy_back_from_buffer on the last layer (cache.py:349-351)
        return logits

    def prefill(self, input_ids):                         # :221-226
        for i in range(math.ceil(input_ids.shape[1] / 128)):
            logits = self.inference(input_ids[:, i * 128:(i + 1) * 128])
        return logits

    def build_retrieval_cache(self, input_ids):           # :228-232
        assert input_ids.shape[-1] == 1
        return self.inference(input_ids, retrieval_cache=self.retrieval_cache)

    def retrieval_tree_inference(self, input_ids, storage_ids, position_ids, attention_mask):
        """:406-425 + tensor_op.TP_Attention_Tree_Retrieval :230-272.  attention_mask dense (n, budget+tree)."""
        rc = self.retrieval_cache

        def attend(i, q, k, v):
            kk, vv = rc.update(k, v, i, storage_ids)
            return attn_sdpa(q, kk, vv, attention_mask)
        return self._forward(input_ids, position_ids, attend)


# ----------------------------------------------------------------------------------------
# SpecTree (utils/SpecTree_TP.py)
# ----------------------------------------------------------------------------------------
class SpecTreeO:
    def __init__(self, engine, grow_map, temperature, top_p, vocab_size, rng, rand):
        self.e, self.gm = engine, grow_map
        self.T, self.top_p, self.V = temperature, top_p, vocab_size
        self.rng = rng
        self.tree_size = grow_map["size"]
        self._own_rand = rand is None                     # rand: (tree_size, V) fp16 uniforms (:90, refreshed :96)
        self.rand = torch.empty((self.tree_size, vocab_size), dtype=torch.float16).uniform_() if rand is None else rand
        self.draft_step = len(grow_map["roots"])
        self.roots = [torch.tensor(x).long() for x in grow_map["roots"]]
        self.branches = grow_map["branches"]
        self.succ = grow_map["Successors"]
        self.tree_size = grow_map["size"]
        self.depth = grow_map["depth"]
        self.tree_mask = additive_tree_mask(grow_map["mask"])                         # :65-67
        rc = engine.retrieval_cache
        self.storage_ids = torch.arange(rc.max_budget, rc.real_budget)                # :76
        self.mask_step, self.sid_step = [], []
        start = 1
        for i in range(self.draft_step - 1):                                          # :80-85
            nb = sum(self.branches[i])
            self.mask_step.append(torch.cat([torch.zeros(nb, rc.max_budget), self.tree_mask[start:start + nb]], dim=-1))
            self.sid_step.append(self.storage_ids[start:start + nb].clone())
            start += nb
        self.mask_first = torch.cat([torch.zeros(1, rc.max_budget), self.tree_mask[0:1]], dim=-1)   # :87
        self.gather = []                                  # offloading_seqouia.py:124-134
        for i in range(self.draft_step - 1):
            mx = max(self.branches[i])
            self.gather.append(torch.cat([torch.arange(b) + j * mx for j, b in enumerate(self.branches[i])]))
        self.draft_logits = torch.zeros(self.tree_size, vocab_size)
        self.verify_tokens = torch.zeros(self.tree_size, dtype=torch.long)
        self.trace = []

    def prefill(self, prefix):                            # :93-101
        self.draft_logits.zero_()
        self.verify_tokens.zero_()
        if self._own_rand:
            self.rand.uniform_()
        self.e.reset()
        self.e.prefill(prefix.unsqueeze(0)[:, :-1])
        logits = self.e.build_retrieval_cache(prefix.unsqueeze(0)[:, -1:])
        return self.rng.sample(R.norm_logits(logits[:, -1, :][0:1], self.T, -1, self.top_p)[0])

    def construct_grow_map(self, next_token):             # :103-145
        S = self.e.kv_cache.seq_len
        self.verify_tokens[0] = next_token
        pos = torch.arange(S, S + 1).unsqueeze(0)
        dl = self.e.retrieval_tree_inference(torch.tensor([[next_token]]), self.storage_ids[0:1], pos, self.mask_first)[0]
        self.draft_logits[0] = dl
        for i in range(self.draft_step - 1):
            idx, nxt = self.roots[i], self.roots[i + 1]
            nb = sum(self.branches[i])
            toks = sample_without_replacement(self.draft_logits[idx], self.rand[idx], max(self.branches[i]), self.T)
            toks = toks[self.gather[i]]
            self.verify_tokens[nxt] = toks
            pos = (self.depth[nxt] + S).unsqueeze(0)
            dl = self.e.retrieval_tree_inference(toks.view(1, nb), self.sid_step[i], pos, self.mask_step[i])[0]
            self.draft_logits[nxt] = dl

    def verify(self):                                     # :168-236
        S = self.e.kv_cache.seq_len
        pos = (self.depth + S).unsqueeze(0)
        mask = torch.cat([torch.zeros(self.tree_size, S), self.tree_mask], dim=-1)
        logits = self.e.inference(self.verify_tokens.unsqueeze(0), position_ids=pos, attention_mask=mask)[0]
        probs = R.norm_logits(logits, self.T, -1, self.top_p)      # == softmax(get_sampling_logits(.)/T), :176-177
        accept_list, next_token, terminal, acc_count = accept_walk(probs, self.draft_logits, self.verify_tokens,
                                                                   self.succ, self.T, self.rng)
        self.trace.append(dict(tokens=self.verify_tokens.clone(), accept_list=list(accept_list), acc_count=acc_count,
                               next_token=next_token, terminal=terminal))
        if terminal:
            return None, acc_count, []
        accept_tokens = self.verify_tokens[accept_list].tolist() + [next_token]
        self.e.kv_cache.gather_kv_incremental(accept_list, S)
        self.e.retrieval_cache.update_graph_cache(self.e.kv_cache)
        self.draft_logits.zero_()
        self.verify_tokens.zero_()
        return next_token, acc_count, accept_tokens


def run_sequoia(spec, prefix, gen_len):
    """The decode loop of test/offloading_seqouia.py:155-185: returns the generated ids and per-step accept counts."""
    next_token = spec.prefill(prefix)
    generated, counts, n = [next_token], [], 0
    while n < gen_len:
        spec.construct_grow_map(next_token)
        next_token, acc, toks = spec.verify()
        if next_token is None:
            break
        generated.extend(toks[1:])
        n += acc
        counts.append(acc)
    return generated, counts
