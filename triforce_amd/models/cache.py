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

    def tail_source(self, layers, prefill):
        """(k, v, first_row) holding the generated-token rows [prefill, seq_len) of `layers` on the device."""
        return self.k[layers], self.v[layers], prefill

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


class OffloadingFlashSimpleCache(Cache):
    """Single-GPU offloading cache (reference cache.py:63-115, entry point test/offloading.py): the whole KV
    lives in pinned host memory and every target forward streams it layer by layer through two device buffers.

    The reference does this synchronously (`.cpu()` of the new rows, then a full-layer H2D on the compute stream,
    :98-104).  Here the live tokens of layer i+1 are prefetched on a copy stream (tf_kv_h2d_async) while layer i
    computes, the new rows go back with tf_kv_d2h_async, and a small device mirror of the generated-token rows
    serves RetrievalCache.update_graph_cache without touching host memory."""

    def __init__(self, model, max_budget=1024, tail_capacity=None) -> None:
        self.seq_len = 0
        self.max_budget = max_budget
        self.layers, self.num_heads, self.head_dim = _geom(model)
        self.hidden_size = model.config.hidden_size
        self.device = model.device
        L, H, T, D = self.layers, self.num_heads, max_budget, self.head_dim
        self.cpu_k = _pinned_zeros(L, H, T, D)
        self.cpu_v = _pinned_zeros(L, H, T, D)
        self.key_cache, self.value_cache = _ref_view(self.cpu_k), _ref_view(self.cpu_v)
        self.buf_k = [torch.zeros(H, T, D, dtype=torch.float16, device=self.device) for _ in range(2)]
        self.buf_v = [torch.zeros(H, T, D, dtype=torch.float16, device=self.device) for _ in range(2)]
        self.key_cache_buffer, self.value_cache_buffer = self.buf_k[0], self.buf_v[0]
        self.load_stream = torch.cuda.Stream(device=self.device)
        self.tail_base = None                       # first token row mirrored in tail_k/v (= prefill length)
        self.tail_capacity = tail_capacity
        self.tail_k = self.tail_v = None
        self._ready = {}

    def print_status(self):
        print("[Offloading Flash Simple Cache] Cached Size:", self.seq_len, "| Budget:", self.max_budget)

    def reset(self):
        self.seq_len = 0
        self.cpu_k.zero_()
        self.cpu_v.zero_()

    def set_tail(self, base, capacity):
        """Mirror rows [base, base+capacity) (the generated tokens) on the device as they are produced."""
        self.tail_base = base
        if self.tail_k is None or self.tail_k.shape[2] < capacity:
            self.tail_k, self.tail_v = _alloc(self.layers, self.num_heads, capacity, self.head_dim, self.device)

    def tail_source(self, layers, prefill):
        assert self.tail_base == prefill, "OffloadingFlashSimpleCache.set_tail(prefill, capacity) must be called first"
        return self.tail_k[layers], self.tail_v[layers], 0

    # -- streaming protocol driven by the model forward ---------------------------------------
    def _fetch(self, layer_idx):
        b = layer_idx % 2
        ops.kv_h2d_async(self.buf_k[b], self.cpu_k[layer_idx], self.seq_len, self.load_stream)
        ops.kv_h2d_async(self.buf_v[b], self.cpu_v[layer_idx], self.seq_len, self.load_stream)
        ev = torch.cuda.Event()
        ev.record(self.load_stream)
        self._ready[layer_idx] = ev

    def begin_forward(self):
        self.load_stream.wait_stream(torch.cuda.current_stream(self.device))
        self._ready = {}
        for i in range(min(2, self.layers)):
            self._fetch(i)

    def layer_kv(self, layer_idx):
        torch.cuda.current_stream(self.device).wait_event(self._ready[layer_idx])
        b = layer_idx % 2
        return self.buf_k[b], self.buf_v[b]

    def append_slot(self, layer_idx, n):
        if self.seq_len + n > self.max_budget:
            raise IndexError(f"OffloadingFlashSimpleCache overflow: {self.seq_len}+{n} > {self.max_budget}")
        return self.seq_len                          # seq_len advances in end_forward (every layer sees the same slot)

    def layer_done(self, layer_idx, slot, n):
        b = layer_idx % 2
        if self.tail_base is not None and slot >= self.tail_base:
            ops.kv_copy_rows_pair(self.buf_k[b].unsqueeze(0), self.buf_v[b].unsqueeze(0), self.tail_k[layer_idx:layer_idx + 1],
                                  self.tail_v[layer_idx:layer_idx + 1], slot, slot - self.tail_base, n)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(self.device))
        self.load_stream.wait_event(done)
        ops.kv_d2h_async(self.cpu_k[layer_idx], self.buf_k[b], slot, n, self.load_stream)
        ops.kv_d2h_async(self.cpu_v[layer_idx], self.buf_v[b], slot, n, self.load_stream)
        if layer_idx + 2 < self.layers:
            self._fetch(layer_idx + 2)
        self._pending = n

    def end_forward(self):
        torch.cuda.current_stream(self.device).wait_stream(self.load_stream)
        self.seq_len += getattr(self, "_pending", 0)
        self._pending = 0


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
        whole = layers == slice(0, self.layers)
        if whole and ops.HOST_PLANS and self.k.is_cuda and type(kv_cache) is FlashSimpleCache:
            # the per-step refresh over all layers: the same tensors every step -> a launch plan (ops.KvCopyPairPlan); only for
            # the resident cache, whose storage never moves (the offloading cache re-allocates its tail mirror)
            plan = getattr(self, "_tail_plan", None)
            if plan is None or plan[0] != (id(kv_cache), kv_cache.k.data_ptr()):
                src_k, src_v, t0 = kv_cache.tail_source(layers, self.prefill)
                plan = self._tail_plan = ((id(kv_cache), kv_cache.k.data_ptr()), ops.KvCopyPairPlan(src_k, src_v, self.k, self.v), t0)
            plan[1](plan[2], self.max_budget - g, g)
            return
        src_k, src_v, t0 = kv_cache.tail_source(layers, self.prefill)
        ops.kv_copy_rows_pair(src_k, src_v, self.k[layers], self.v[layers], t0, self.max_budget - g, g)

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
        ops.kv_shift_rows_pair(self.k, self.v, self.seq_len - size_keep, self.start_size, size_keep)
        self.seq_len = self.start_size + self.recent_size - incoming

    def evict_for_spec(self, current_seq_len):
        if ops.HOST_PLANS and self.k.is_cuda:
            plan = getattr(self, "_shift_plan", None)
            if plan is None:
                plan = self._shift_plan = ops.KvShiftPairPlan(self.k, self.v)
            plan(current_seq_len - self.recent_size, self.start_size, self.recent_size)
            return
        ops.kv_shift_rows_pair(self.k, self.v, current_seq_len - self.recent_size, self.start_size, self.recent_size)


# =============================================================================================
# Tensor-parallel / offloading caches (reference cache.py:268-383, :485-584)
# =============================================================================================
def _pin(t):
    return t.pin_memory() if torch.cuda.is_available() else t


def _pinned_zeros(*shape):
    """fp16 zeros in pinned host memory, allocated pinned from the start (t.pin_memory() would hold a pageable AND a
    pinned copy of a 50 GB offloaded cache for a moment)."""
    return torch.zeros(*shape, dtype=torch.float16, pin_memory=torch.cuda.is_available())


class DistributedSimpleCache(Cache):
    """This rank's heads of the full KV cache: layers < on_chip_layers live in HBM, the rest in pinned host
    memory and are streamed through two DistributedKVCacheBuffer's (reference cache.py:268-351).
    Unlike the reference, on_chip_layers == num_layers (nothing offloaded — the natural setting with
    288 GB of HBM) is allowed."""

    def __init__(self, config, max_budget=1024, device=None, on_chip_layers=0, ssl=0):
        self.config = config
        self.world_size, self.local_rank = config.world_size, config.local_rank
        self.device = torch.device(device)
        self.max_budget = max_budget
        self.hidden_size = config.hidden_size
        self.num_heads = config.num_key_value_heads // self.world_size
        self.head_dim = config.hidden_size // config.num_attention_heads
        self.layers = config.num_hidden_layers
        self.on_chip_layers = min(on_chip_layers, self.layers)
        self.seq_len = 0
        n_on, n_off = self.on_chip_layers, self.layers - self.on_chip_layers
        self.k, self.v = _alloc(n_on, self.num_heads, max_budget, self.head_dim, self.device)
        self.cpu_k = _pinned_zeros(n_off, self.num_heads, max_budget, self.head_dim)
        self.cpu_v = _pinned_zeros(n_off, self.num_heads, max_budget, self.head_dim)
        self.key_cache, self.value_cache = _ref_view(self.k), _ref_view(self.v)
        self.cpu_key_cache, self.cpu_value_cache = _ref_view(self.cpu_k), _ref_view(self.cpu_v)

    def print_status(self):
        print("Cached Size:", self.seq_len, "| Max Budget:", self.max_budget)

    def reset(self):
        self.seq_len = 0
        self.cpu_k.zero_()
        self.cpu_v.zero_()
        self.k.zero_()
        self.v.zero_()

    def normal_(self, seq_len=1024 * 127):
        """The reference's own synthetic filler (cache.py:303-308)."""
        self.seq_len = seq_len
        for t in (self.cpu_k, self.cpu_v, self.k, self.v):
            t.normal_()

    def layer_kv(self, layer_idx):
        assert layer_idx < self.on_chip_layers, (layer_idx, self.on_chip_layers)
        return self.k[layer_idx], self.v[layer_idx]

    def host_layer_kv(self, layer_idx):
        i = layer_idx - self.on_chip_layers
        assert 0 <= i, (layer_idx, self.on_chip_layers)
        return self.cpu_k[i], self.cpu_v[i]

    def gather_kv_incremental(self, indices, offset):
        """Sequoia: keep only the accepted tree nodes — rows offset+indices[j] -> offset+j of every layer, then
        seq_len = offset + len(indices) (reference cache.py:333-343).  On-chip layers: one tf_kv_gather_rows
        launch; host-resident layers: pinned-memory row copies; the device mirror of the generated rows kept by
        the retrieval cache (``tail_mirror``) is compacted the same way."""
        idx = [int(i) for i in indices]
        assert all(b > a for a, b in zip(idx, idx[1:])), "accept list must be strictly increasing"
        n = len(idx)
        if n and idx != list(range(n)):
            dev_idx = torch.tensor(idx, dtype=torch.int32, device=self.device)
            if self.on_chip_layers > 0:
                ops.kv_gather_rows(self.k, self.v, offset, dev_idx, max_index=idx[-1])
            if self.on_chip_layers < self.layers:
                src = torch.tensor([offset + i for i in idx], dtype=torch.long)
                for t in (self.cpu_k, self.cpu_v):
                    t[:, :, offset:offset + n] = t[:, :, src]
            tm = getattr(self, "tail_mirror", None)
            if tm is not None and tm.tail_k is not None:
                ops.kv_gather_rows(tm.tail_k, tm.tail_v, offset - tm.prefill, dev_idx, max_index=idx[-1])
        self.seq_len = offset + n


class DistributedKVCacheBuffer:
    """Device staging buffer for one offloaded layer (reference cache.py:353-383)."""

    def __init__(self, config, max_budget=1024, device=None) -> None:
        self.config = config
        self.max_budget = max_budget
        self.device = torch.device(device)
        self.num_heads = config.num_key_value_heads // config.world_size
        self.head_dim = config.hidden_size // config.num_attention_heads
        self.k = torch.zeros(self.num_heads, max_budget, self.head_dim, dtype=torch.float16, device=self.device)
        self.v = torch.zeros(self.num_heads, max_budget, self.head_dim, dtype=torch.float16, device=self.device)
        self.key_cache, self.value_cache = self.k.permute(1, 0, 2).unsqueeze(0), self.v.permute(1, 0, 2).unsqueeze(0)
        self.seq_len = 0

    def copy_kv(self, kv_cache, layer_idx, stream):
        """Stream the live tokens of one offloaded layer host -> device on `stream` (cache.py:372-376)."""
        hk, hv = kv_cache.host_layer_kv(layer_idx)
        ops.kv_h2d_async(self.k, hk, kv_cache.seq_len, stream)
        ops.kv_h2d_async(self.v, hv, kv_cache.seq_len, stream)
        self.seq_len = kv_cache.seq_len

    def copy_back(self, kv_cache, layer_idx, t0, n, stream):
        """Write the n new tokens back to the pinned host cache (cache.py:345-351)."""
        hk, hv = kv_cache.host_layer_kv(layer_idx)
        ops.kv_d2h_async(hk, self.k, t0, n, stream)
        ops.kv_d2h_async(hv, self.v, t0, n, stream)


class DistributedRetrievalCache:
    """Per-rank retrieval cache, always in HBM (reference cache.py:485-584).  Differences from the
    single-GPU class that the reference has and we keep: reset() clears init_graph, and building twice
    without a reset raises (cache.py:519-520,577-580)."""

    def __init__(self, config, max_budget=1024, device=None, prefill=1024, chunk_size=8, gamma=6) -> None:
        self.config = config
        self.world_size, self.local_rank = config.world_size, config.local_rank
        self.device = torch.device(device)
        self.hidden_size = config.hidden_size
        self.num_heads = config.num_key_value_heads // self.world_size
        self.head_dim = config.hidden_size // config.num_attention_heads
        self.layers = config.num_hidden_layers
        self.chunk_size, self.prefill, self.gamma, self.max_budget = chunk_size, prefill, gamma, max_budget
        self.chunks = prefill // chunk_size
        self.select_sets = max_budget // chunk_size
        assert prefill % self.chunk_size == 0, f"prefill should be multiple of chunk_size, got {prefill} % {self.chunk_size}"
        assert max_budget % self.chunk_size == 0, f"max_budget should be multiple of chunk_size, got {max_budget} % {self.chunk_size}"
        self.real_budget = max_budget + gamma + 1
        self.init_graph = False
        self.k, self.v = _alloc(self.layers, self.num_heads, self.real_budget, self.head_dim, self.device)
        self.key_cache, self.value_cache = _ref_view(self.k), _ref_view(self.v)
        # device-side copy of the generated-token KV of EVERY layer ([prefill, seq_len) rows), filled as the
        # target forward produces them, so refreshing the retrieval tail never has to touch host memory
        self.tail_k = self.tail_v = None
        self.last_scores = [None] * self.layers
        self.last_idx = [None] * self.layers

    def print_status(self):
        print("Budget:", self.max_budget, " | Real Budget:", self.real_budget, " | PreFill:", self.prefill, " | Chunk Size:",
              self.chunk_size, " | Chunks:", self.chunks, " | Select Sets:", self.select_sets)

    def layer_kv(self, layer_idx):
        return self.k[layer_idx], self.v[layer_idx]

    @property
    def spec_slot(self):
        return self.real_budget - self.gamma - 1

    def init_graph_cache(self, kv_view, query_states, layer_idx):
        """kv_view: (k_layer, v_layer) head-major views holding this layer's prefix (HBM cache or staging buffer)."""
        if self.init_graph:
            raise ValueError("Graph is already initialized")
        q = query_states.reshape(-1, self.num_heads, self.head_dim)
        assert 1 == q.shape[0], "query_states should be 1 for init"
        src_k, src_v = kv_view
        scores = ops.retrieval_score(src_k, q[0].contiguous(), self.chunks, self.chunk_size)
        idx = ops.retrieval_topk(scores, self.select_sets)
        ops.retrieval_gather(src_k, src_v, idx, self.k[layer_idx], self.v[layer_idx], self.chunk_size)
        self.last_scores[layer_idx], self.last_idx[layer_idx] = scores, idx
        if layer_idx == self.layers - 1:
            self.init_graph = True

    def update(self, key_states, value_states, layer_idx):
        k, v = _rows(key_states), _rows(value_states)
        s = self.spec_slot
        self.k[layer_idx, :, s:] = k.permute(1, 0, 2)
        self.v[layer_idx, :, s:] = v.permute(1, 0, 2)
        return self.key_cache[layer_idx][:, :self.real_budget], self.value_cache[layer_idx][:, :self.real_budget]

    def ensure_tail(self, capacity):
        if self.tail_k is None or self.tail_k.shape[2] < capacity:
            self.tail_k, self.tail_v = _alloc(self.layers, self.num_heads, capacity, self.head_dim, self.device)

    def update_graph_cache(self, kv_cache=None):
        """Copy the generated tokens' KV [prefill, seq_len) of all layers into slots [B-g, B)
        (cache.py:566-575).  The reference reads the offloaded layers back from pinned host memory here;
        we keep those rows in `tail_k/v` on the device (written during the target forward)."""
        g = kv_cache.seq_len - self.prefill
        if g <= 0:
            return
        if g > self.max_budget:
            raise IndexError(f"generated tail ({g}) exceeds the retrieval budget ({self.max_budget})")
        ops.kv_copy_rows_pair(self.tail_k, self.tail_v, self.k, self.v, 0, self.max_budget - g, g)

    def reset(self):
        self.k.zero_()
        self.v.zero_()
        self.init_graph = False

    def normal_(self):
        self.k.normal_()
        self.v.normal_()


class DistributedRetrievalCache_Seqouia(DistributedRetrievalCache):
    """Retrieval cache of the Sequoia tree path (reference cache.py:385-483; the reference's spelling of the
    class name is kept).  Same chunk-scored build as DistributedRetrievalCache; the speculative region holds one
    KV row per TREE NODE — slot max_budget + node id, written through ``storage_ids`` (:459-466) — instead of the
    gamma+1 chain slots, so real_budget = max_budget + tree_size."""

    def __init__(self, config, max_budget=1024, device=None, prefill=1024, chunk_size=8, tree_size=128) -> None:
        super().__init__(config, max_budget=max_budget, device=device, prefill=prefill, chunk_size=chunk_size,
                         gamma=tree_size - 1)
        self.tree_size = tree_size
        assert self.real_budget == max_budget + tree_size

    def update(self, key_states, value_states, layer_idx, storage_ids):
        k, v = _rows(key_states), _rows(value_states)
        input_length = len(storage_ids)
        assert input_length == k.shape[0] and input_length == v.shape[0]
        ids = torch.as_tensor(storage_ids, device=self.k.device, dtype=torch.long)
        self.k[layer_idx].index_copy_(1, ids, k.permute(1, 0, 2).contiguous())
        self.v[layer_idx].index_copy_(1, ids, v.permute(1, 0, 2).contiguous())
        return self.key_cache[layer_idx], self.value_cache[layer_idx]
