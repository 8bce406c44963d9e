"""KV caches of the TriForce path — host-side mirror of the reference's models/cache.py
(FlashSimpleCache :20-61, RetrievalCache :117-198, StreamingLLMEvictionCache :200-265) with the
same constructors, attributes and methods, backed by HIP kernels.

Physical layout is HEAD-MAJOR ``[L][H][T][D]`` fp16 (``.k`` / ``.v``): one head's keys are
contiguous, so a split-KV wavefront streams one contiguous 256-B-per-key segment and the retrieval
scorer reads whole 2-KiB chunks.  ``key_cache`` / ``value_cache`` keep the reference's shape
``[L, 1, T, H, D]`` as permuted views of the same storage (nothing outside models/ indexes them,
SURVEY §8b B3).

Reference quirks kept on purpose: RetrievalCache.reset() does not clear
``init_graph``; StreamingLLMEvictionCache.reset() does not reset ``seq_len``.
"""
from typing import Tuple

import torch

from .. import ops


class Cache:
    """Base class (reference models/cache.py:6-17)."""

    def update(self, key_states, value_states, layer_idx) -> Tuple[torch.Tensor, torch.Tensor]:
        raise NotImplementedError("Make sure to implement `update` in a subclass.")


def _geom(model):
    cfg = model.config
    heads = cfg.num_key_value_heads
    return cfg.num_hidden_layers, heads, cfg.hidden_size // cfg.num_attention_heads


def _alloc(L, H, T, D, device):
    k = torch.zeros(L, H, T, D, dtype=torch.float16, device=device)
    v = torch.zeros(L, H, T, D, dtype=torch.float16, device=device)
    return k, v


def _ref_view(t):
    # (L,H,T,D) storage -> the reference's [L,1,T,H,D] shape
    return t.permute(0, 2, 1, 3).unsqueeze(1)


def _rows(x):
    # accepts the reference's (1,n,H,D) or (n,H,D); returns (n,H,D)
    return x[0] if x.dim() == 4 else x


class FlashSimpleCache(Cache):
    """Full KV cache of the target model (reference cache.py:20-61)."""

    def __init__(self, model, max_budget=1024) -> None:
        self.seq_len = 0
        self.max_budget = max_budget
        self.layers, self.num_heads, self.head_dim = _geom(model)
        self.hidden_size = model.config.hidden_size
        self.device = model.device
        self.k, self.v = _alloc(self.layers, self.num_heads, max_budget, self.head_dim, self.device)
        self.key_cache, self.value_cache = _ref_view(self.k), _ref_view(self.v)
        self.scores = []

    def print_status(self):
        print("[Full Cache] Cached:", self.seq_len, "| Budget:", self.max_budget)

    def reset(self):
        self.seq_len = 0
        self.k.zero_()
        self.v.zero_()

    def layer_kv(self, layer_idx):
        return self.k[layer_idx], self.v[layer_idx]

    def append_slot(self, layer_idx, n):
        """Slot where the fused RoPE+append kernel writes this layer's n new rows; seq_len advances
        after the last layer exactly like update() (cache.py:58-59)."""
        if self.seq_len + n > self.max_budget:
            raise IndexError(f"FlashSimpleCache overflow: {self.seq_len}+{n} > {self.max_budget}")
        slot = self.seq_len
        if layer_idx == self.layers - 1:
            self.seq_len += n
        return slot

    def update(self, key_states, value_states, layer_idx):
        k, v = _rows(key_states), _rows(value_states)
        n = k.shape[0]
        s = self.seq_len
        self.k[layer_idx, :, s:s + n] = k.permute(1, 0, 2)
        self.v[layer_idx, :, s:s + n] = v.permute(1, 0, 2)
        key = self.key_cache[layer_idx][:, :s + n]
        value = self.value_cache[layer_idx][:, :s + n]
        if layer_idx == self.layers - 1:
            self.seq_len += n
        return key, value


class RetrievalCache(Cache):
    """Per-head retrieval cache (reference cache.py:117-198): chunk-mean scoring, top-k, gather."""

    def __init__(self, model, max_budget=1024, prefill=1024, chunk_size=8, gamma=6) -> None:
        self.chunk_size = chunk_size
        self.prefill = prefill
        self.chunks = prefill // self.chunk_size
        self.select_sets = max_budget // self.chunk_size
        self.gamma = gamma
        self.max_budget = max_budget
        assert prefill % self.chunk_size == 0, f"prefill should be multiple of chunk_size, got {prefill} % {self.chunk_size}"
        assert max_budget % self.chunk_size == 0, f"max_budget should be multiple of chunk_size, got {max_budget} % {self.chunk_size}"
        self.real_budget = max_budget + gamma + 1
        self.layers, self.num_heads, self.head_dim = _geom(model)
        self.hidden_size = model.config.hidden_size
        self.device = model.device
        self.k, self.v = _alloc(self.layers, self.num_heads, self.real_budget, self.head_dim, self.device)
        self.key_cache, self.value_cache = _ref_view(self.k), _ref_view(self.v)
        self.init_graph = False
        self.last_scores = [None] * self.layers      # (H,C) fp16 / (H,sets) int32 of the last build, for parity checks
        self.last_idx = [None] * self.layers

    def print_status(self):
        print("[Retrieval Cache] Budget:", self.max_budget, " | PreFill:", self.prefill, " | Chunk Size:", self.chunk_size,
              " | Chunks:", self.chunks, " | Select Sets:", self.select_sets)

    def layer_kv(self, layer_idx):
        return self.k[layer_idx], self.v[layer_idx]

    @property
    def spec_slot(self):
        return self.real_budget - self.gamma - 1

    def init_graph_cache(self, kv_cache, query_states, layer_idx):
        q = query_states.reshape(-1, self.num_heads, self.head_dim)
        assert 1 == q.shape[0], "query_states should be 1 for init"
        src_k, src_v = kv_cache.layer_kv(layer_idx)
        scores = ops.retrieval_score(src_k, q[0].contiguous(), self.chunks, self.chunk_size)
        idx = ops.retrieval_topk(scores, self.select_sets)
        ops.retrieval_gather(src_k, src_v, idx, self.k[layer_idx], self.v[layer_idx], self.chunk_size)
        self.last_scores[layer_idx], self.last_idx[layer_idx] = scores, idx
        if layer_idx == self.layers - 1:
            self.init_graph = True

    def _copy_tail(self, kv_cache, layers):
        g = kv_cache.seq_len - self.prefill
        if g > self.max_budget:
            raise IndexError(f"generated tail ({g}) exceeds the retrieval budget ({self.max_budget})")
        ops.kv_copy_rows(kv_cache.k[layers], self.k[layers], self.prefill, self.max_budget - g, g)
        ops.kv_copy_rows(kv_cache.v[layers], self.v[layers], self.prefill, self.max_budget - g, g)

    def update_graph_cache(self, kv_cache=None):
        self._copy_tail(kv_cache, slice(0, self.layers))

    def update_graph_cache_retrieval(self, kv_cache, query_states, layer_idx):
        self.init_graph_cache(kv_cache, query_states, layer_idx)
        self._copy_tail(kv_cache, slice(layer_idx, layer_idx + 1))

    def update(self, new_k_cache, new_v_cache, layer_idx):
        k, v = _rows(new_k_cache), _rows(new_v_cache)
        s = self.spec_slot
        self.k[layer_idx, :, s:] = k.permute(1, 0, 2)
        self.v[layer_idx, :, s:] = v.permute(1, 0, 2)
        return self.key_cache[layer_idx][:, :self.real_budget], self.value_cache[layer_idx][:, :self.real_budget]

    def reset(self):
        self.k.zero_()
        self.v.zero_()


class StreamingLLMEvictionCache(Cache):
    """Sink + sliding-window cache of the 68M draft (reference cache.py:200-265); keys un-rotated."""

    def __init__(self, model, gamma=6, start_size=16, recent_size=496) -> None:
        self.gamma = gamma
        self.start_size = start_size
        self.recent_size = recent_size
        self.real_budget = self.start_size + self.recent_size + self.gamma + 1 + 1 + 1
        self.seq_len = 0  # just for prefill usage
        self.layers, self.num_heads, self.head_dim = _geom(model)
        self.hidden_size = model.config.hidden_size
        self.device = model.device
        self.k, self.v = _alloc(self.layers, self.num_heads, self.real_budget, self.head_dim, self.device)
        self.key_cache, self.value_cache = _ref_view(self.k), _ref_view(self.v)

    def print_status(self):
        print("[StreamingLLM Cache] Start Size:", self.start_size, "| Recent Size:", self.recent_size, "| Gamma:", self.gamma,
              "| Real Budget:", self.real_budget, "| Cached:", self.seq_len)

    def layer_kv(self, layer_idx):
        return self.k[layer_idx], self.v[layer_idx]

    @property
    def spec_slot(self):
        return self.real_budget - self.gamma - 3

    def append_slot(self, layer_idx, n):
        assert self.seq_len + n <= self.start_size + self.recent_size
        slot = self.seq_len
        if layer_idx == self.layers - 1:
            self.seq_len += n
        return slot

    def update(self, key_states, value_states, layer_idx):
        k, v = _rows(key_states), _rows(value_states)
        n = k.shape[0]
        assert self.seq_len + n <= self.start_size + self.recent_size
        s = self.seq_len
        self.k[layer_idx, :, s:s + n] = k.permute(1, 0, 2)
        self.v[layer_idx, :, s:s + n] = v.permute(1, 0, 2)
        key = self.key_cache[layer_idx][:, :s + n]
        value = self.value_cache[layer_idx][:, :s + n]
        if layer_idx == self.layers - 1:
            self.seq_len += n
        return key, value

    def spec_update(self, new_k_cache, new_v_cache, layer_idx, gamma_offset=0):
        k, v = _rows(new_k_cache), _rows(new_v_cache)
        start = self.spec_slot
        end = start + k.shape[0]
        self.k[layer_idx, :, start:end] = k.permute(1, 0, 2)
        self.v[layer_idx, :, start:end] = v.permute(1, 0, 2)
        return self.key_cache[layer_idx][:, :end], self.value_cache[layer_idx][:, :end]

    def reset(self):
        self.k.zero_()
        self.v.zero_()

    def evict_prefill(self, incoming):
        if self.seq_len + incoming <= self.start_size + self.recent_size:
            return
        size_keep = self.recent_size - incoming
        ops.kv_shift_rows(self.k, self.seq_len - size_keep, self.start_size, size_keep)
        ops.kv_shift_rows(self.v, self.seq_len - size_keep, self.start_size, size_keep)
        self.seq_len = self.start_size + self.recent_size - incoming

    def evict_for_spec(self, current_seq_len):
        ops.kv_shift_rows(self.k, current_seq_len - self.recent_size, self.start_size, self.recent_size)
        ops.kv_shift_rows(self.v, current_seq_len - self.recent_size, self.start_size, self.recent_size)
