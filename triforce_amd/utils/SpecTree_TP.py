"""Sequoia static-tree speculation on top of the retrieval cache — host-side mirror of the reference's
utils/SpecTree_TP.py (SpecTree :31-236): same constructor keywords and the same three calls the entry script
makes (prefill / construct_grow_map / verify, test/offloading_seqouia.py:160-166).

Algorithm (unchanged): the target weights running over the retrieval cache grow a static tree level by level —
the children of a node are drawn WITHOUT replacement from softmax(draft_logits / T) by the exponential-race
trick ``topk(log(u) / q)`` (offloading_seqouia.py:29-39) — then the target verifies all ``tree_size`` nodes in
ONE forward against the full KV cache with tree attention, walks the tree accepting children by the multi-round
residual rule (:147-167), and the accepted nodes' KV rows are compacted (gather_kv_incremental).

Execution differences:
  * tree attention reads 512-bit mask rows for the tree columns only (models/TP_llama_tree.py) instead of dense
    (q x (S+tree)) additive masks;
  * the accept walk, the residual updates and the final sample run in ONE single-workgroup kernel
    (tf_tree_accept) with an explicit uniform stream — the reference syncs the host per examined child and
    broadcasts five tensors with barriers per step (:205-223); here one 64-word record is read (and, under TP,
    broadcast once from rank 0);
  * token buffers stay on the device; the only host reads are that record per verify.
"""
import torch
import torch.distributed as dist

from .. import ops
from ..models.TP_llama import TreeMask
from .sampling import UniformSource, norm_logits, sample
from .tree import successors_csr


def sampling_without_replacement(num_samples, temperature=0.6):
    """offloading_seqouia.py:29-39 `create_sampling_callable`: (logits (n,V) fp32, rand (n,V) fp16) -> n*num_samples
    token ids, per row the num_samples largest log(u)/q — a draw without replacement proportional to q.  On the
    device this is one kernel (tf_sample_without_replacement); the torch formulation below is the host-side
    statement of the same arithmetic."""
    def run(sampling_logits, static_rand):
        if sampling_logits.is_cuda and sampling_logits.shape[-1] <= ops.TOPP_MAX_VOCAB and num_samples <= 16:
            return ops.sample_without_replacement(sampling_logits, static_rand, num_samples, temperature)
        q = torch.softmax(sampling_logits / temperature, dim=-1)
        return (static_rand.log() / q).topk(k=num_samples).indices.flatten()
    return run


def get_sampling_logits(logits, top_p, T, replicate=False):
    """SpecTree_TP.py:9-21 (kept for API parity; verify() uses the fused tf_topp_probs, which computes
    softmax(get_sampling_logits(logits) / T) in one kernel)."""
    if replicate:
        logits = logits.clone()
    if top_p < 1.0:
        sorted_logits, sorted_indices = torch.sort(logits, descending=True, stable=True)
        cumulative_probs = torch.cumsum(torch.softmax(sorted_logits / T, dim=-1), dim=-1)
        filt = cumulative_probs > top_p
        filt[..., 1:] = filt[..., :-1].clone()
        filt[..., 0] = 0
        logits[filt.scatter(-1, sorted_indices, filt)] = float("-inf")
    return logits


def _bcast(t):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=0)


def sample_dist(probs, rng=None):
    """Rank 0's sample is everyone's (SpecTree_TP.py:23-29)."""
    tok = sample(probs, rng=rng)
    _bcast(tok)
    return tok


def _worst_walk(successors):
    """(most children any root-to-leaf walk can examine, deepest path) of a tree given as per-node successor lists."""
    best_c, best_d = [0] * len(successors), [0] * len(successors)
    for node in range(len(successors) - 1, -1, -1):          # children have larger ids than their parent
        kids = successors[node]
        if kids:
            best_c[node] = len(kids) + max(best_c[k] for k in kids)
            best_d[node] = 1 + max(best_d[k] for k in kids)
    return best_c[0], best_d[0]


class SpecTree:
    def __init__(self, engine, temperature: float = 0.6, top_p: float = 0.9, max_length=256, vocab_size=32000,
                 grow_map=None, residual_graph=None, sampling_callables=None, sample_gather_indices=None,
                 tokenizer=None, rng=None, rand_values=None) -> None:
        self.graph_engine = engine
        self.temperature, self.top_p = temperature, top_p
        self.residual_graph = residual_graph            # relu(p-q)/sum — computed inside tf_tree_accept
        self.tokenizer = tokenizer
        self.device = engine.device
        self.dtype = torch.float16
        self.vocab_size = vocab_size
        self.rng = rng or UniformSource(self.device, seed=1)
        self.rand_values = rand_values                   # parity tests inject the (tree_size, V) uniform table

        self.grow_map = grow_map
        self.draft_step = len(grow_map["roots"])
        self.tree_size = int(grow_map["size"])
        self.Successors = grow_map["Successors"]
        self.branches = grow_map["branches"]
        self.level_start = []                            # first node id of level i+1 (levels hold consecutive ids)
        start = 1
        for i in range(self.draft_step - 1):
            roots_next = grow_map["roots"][i + 1]
            assert roots_next == list(range(start, start + sum(self.branches[i]))), "levels must hold consecutive ids"
            self.level_start.append(start)
            start += sum(self.branches[i])
        assert start == self.tree_size
        # tf_tree_accept walks one root-to-leaf path: it reads one uniform per examined child out of a MAX_TAKE-wide
        # window and records at most TREE_ACCEPT_MAX_PATH accepted nodes — refuse trees that could exceed either
        worst_children, worst_depth = _worst_walk(self.Successors)
        if worst_children + 1 > UniformSource.MAX_TAKE or worst_depth + 1 > ops.TREE_ACCEPT_MAX_PATH:
            raise ValueError(f"grow map too deep / wide for the device-side accept walk: a path can examine "
                             f"{worst_children} children over {worst_depth} levels (limits {UniformSource.MAX_TAKE - 1} / "
                             f"{ops.TREE_ACCEPT_MAX_PATH - 1})")
        self.sampling_callables = sampling_callables or {
            i: sampling_without_replacement(max(self.branches[i]), temperature) for i in range(self.draft_step - 1)}
        if sample_gather_indices is None:                # offloading_seqouia.py:124-134
            sample_gather_indices = {}
            for i in range(self.draft_step - 1):
                mx = max(self.branches[i])
                sample_gather_indices[i] = torch.cat([torch.arange(b, dtype=torch.long) + j * mx
                                                      for j, b in enumerate(self.branches[i])]).to(self.device)
        self.sample_gather_indices = sample_gather_indices

        rc = engine.retrieval_cache
        assert rc.real_budget - rc.max_budget == self.tree_size, "retrieval cache was built for another tree size"
        self.storage0 = rc.max_budget                    # slot of node 0 (SpecTree_TP.py:76)
        self.mask_bits = ops.pack_tree_mask(grow_map["mask"].to(self.device))        # (N, N/32) int32
        self.depth = grow_map["depth"].to(self.device)
        self.succ_off, self.succ = successors_csr(self.Successors, self.device)

        N, V = self.tree_size, vocab_size
        self.draft_logits = torch.zeros((N, V), dtype=torch.float32, device=self.device)
        self.rand = torch.empty((N, V), dtype=self.dtype, device=self.device)
        self._refresh_rand()
        self.verify_tokens = torch.zeros(N, dtype=torch.long, device=self.device)
        self.record = torch.zeros(ops.TREE_ACCEPT_OUT, dtype=torch.int64, device=self.device)

    def _refresh_rand(self):
        if self.rand_values is not None:
            self.rand.copy_(torch.as_tensor(self.rand_values).to(self.dtype))
        else:
            self.rand.uniform_()

    # ---------------------------------------------------------------------------------------
    @torch.inference_mode()
    def prefill(self, prefix: torch.LongTensor):
        """SpecTree_TP.py:93-101."""
        self.draft_logits.zero_()
        self.verify_tokens.zero_()
        self._refresh_rand()
        eng = self.graph_engine
        eng.reset()
        if prefix.numel() != eng.prefill_len:
            raise ValueError(f"prompt length {prefix.numel()} != the engine's prefill length {eng.prefill_len}")
        eng.prefill(input_ids=prefix.unsqueeze(0)[:, :-1])
        logits = eng.build_retrieval_cache(input_ids=prefix.unsqueeze(0)[:, -1:])
        return sample_dist(norm_logits(logits[:, -1, :], temperature=self.temperature, top_k=-1, top_p=self.top_p),
                           rng=self.rng)

    @torch.inference_mode()
    def capture_grow_graph(self):
        """Capture the whole tree growth (all levels: sampling without replacement + the retrieval-cache forward of
        every level) as ONE hipGraph.  Every shape is static — the tree is; only the root token and the cache length
        change between steps, and both are read from device buffers (``verify_tokens[0]``, ``_seq_base``).  The
        reference launches ~16 x 32 x 10 kernels per step eagerly.  Single-rank only unless the engine's
        whole-forward RCCL capture was validated (the level forwards contain the all-reduces)."""
        from .graph_infer import _capture
        eng = self.graph_engine
        if eng.world_size > 1 and getattr(eng, "graph_form", "eager") != "whole":
            return False
        self._seq_base = torch.zeros(1, dtype=torch.long, device=self.device)
        pool = getattr(eng, "_mempool", None) or torch.cuda.graphs.graph_pool_handle()
        self._grow_graph, _ = _capture(lambda: self._grow(self._seq_base), (), pool, 2)
        self.draft_logits.zero_()
        self.verify_tokens.zero_()
        eng.retrieval_cache.k[:, :, self.storage0:].zero_()          # the warm-up runs wrote tree slots
        eng.retrieval_cache.v[:, :, self.storage0:].zero_()
        return True

    @torch.inference_mode()
    def construct_grow_map(self, next_token):
        """Grow the tree level by level with the retrieval-cache model (SpecTree_TP.py:103-145)."""
        self.verify_tokens[0:1] = next_token.reshape(-1)[:1]
        if getattr(self, "_grow_graph", None) is not None:
            self._seq_base.fill_(self.graph_engine.kv_cache.seq_len)
            self._grow_graph.replay()
            return
        self._grow(self.graph_engine.kv_cache.seq_len)

    def _grow(self, S):
        """S: cache length — a Python int (eager) or a 1-element device tensor (captured)."""
        eng = self.graph_engine
        logits = eng.retrieval_tree_inference(
            input_ids=self.verify_tokens[0:1].unsqueeze(0), storage_ids=range(self.storage0, self.storage0 + 1),
            position_ids=self.depth[0:1].unsqueeze(0) + S, attention_mask=TreeMask(self.mask_bits, 0))[0]
        self.draft_logits[0:1] = logits
        lo = 0                                            # nodes of level i are [lo, hi)
        hi = 1
        for i in range(self.draft_step - 1):
            total = sum(self.branches[i])
            start = self.level_start[i]
            toks = self.sampling_callables[i](self.draft_logits[lo:hi], self.rand[lo:hi])
            toks = toks[self.sample_gather_indices[i]]
            self.verify_tokens[start:start + total] = toks
            logits = eng.retrieval_tree_inference(
                input_ids=toks.view(1, total), storage_ids=range(self.storage0 + start, self.storage0 + start + total),
                position_ids=(self.depth[start:start + total] + S).unsqueeze(0),
                attention_mask=TreeMask(self.mask_bits, start))[0]
            self.draft_logits[start:start + total] = logits
            lo, hi = start, start + total

    @torch.inference_mode()
    def verify(self):
        """One target forward over all tree nodes + the device-side accept walk (SpecTree_TP.py:168-236).
        Returns (next_token (1,), acc_count, accept_tokens) or (None, acc_count, []) at a terminal."""
        eng = self.graph_engine
        offset = eng.kv_cache.seq_len
        position_ids = (self.depth + offset).unsqueeze(0)
        logits = eng.inference(input_ids=self.verify_tokens.unsqueeze(0), position_ids=position_ids,
                               attention_mask=TreeMask(self.mask_bits, 0))[0]
        # softmax(get_sampling_logits(logits, top_p, T) / T)  (:176-177) — one fused kernel
        self.target_logits = norm_logits(logits, temperature=self.temperature, top_k=-1, top_p=self.top_p)
        ops.tree_accept(self.target_logits, self.draft_logits, self.verify_tokens, self.succ_off, self.succ,
                        self.rng.take(UniformSource.MAX_TAKE), self.temperature, self.record)
        _bcast(self.record)                               # rank 0 decides (:205-223: five broadcasts + barriers)
        rec = self.record.tolist()                        # the one host sync of the step
        nacc, next_tok, terminal, consumed = rec[0], rec[1], bool(rec[2]), rec[3]
        self.rng.advance(consumed)
        accept_list = rec[4:4 + nacc]
        if terminal:
            return None, nacc - 1, []
        acc_count = nacc                                  # accepted nodes + the residual / bonus token
        idx = torch.tensor(accept_list, dtype=torch.long, device=self.device)
        next_token = torch.tensor([next_tok], dtype=torch.long, device=self.device)
        accept_tokens = torch.cat([self.verify_tokens[idx], next_token], dim=-1)
        eng.kv_cache.gather_kv_incremental(accept_list, offset)
        eng.retrieval_cache.update_graph_cache(eng.kv_cache)
        self.draft_logits.zero_()
        self.verify_tokens.zero_()
        return next_token, acc_count, accept_tokens
