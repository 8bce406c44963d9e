"""Single-GPU engine — mirror of the reference's utils/graph_infer.py: InferenceEngine (:14-72)
routes prefill vs verify vs draft; GraphInferenceEngine (:129-194) captures gamma+3 draft hipGraphs
(one per gamma_offset) and one retrieval-verify hipGraph and replays them.

On ROCm ``torch.cuda.CUDAGraph`` is a hipGraph; our kernels are enqueued on torch's current stream
through the C ABI, so they are captured together with the hipBLASLt GEMMs.
"""
import gc
import math
import os
from typing import Optional

import torch

from .. import ops
from .sampling import norm_logits


# Target prefill: the reference feeds 128 tokens per forward (graph_infer.py:30-37, TP_llama.py:246-250), which at
# 128 rows leaves every GEMM bound by re-streaming the 13 GB of weights (976 times for a 125K prompt).  On the device
# the chunk is 4096 rows — the most tf_attn_prefill takes in one launch; same causal attention, 1/32 of the weight
# traffic, and under KV offloading 1/32 of the host->device re-streams (125K prompt, 7B, target model only: round 1
# 15.4 s at 128 rows, 12.6 s at 512, 11.75 s at 1024; round 3, with the faster prefill attention, 6.69 s at 1024, 6.63 s
# at 2048, 6.38 s at 4096: profiles/r03_prefill_chunk.jsonl).  The returned logits keep the
# reference's shape (the rows of ITS last 128-token chunk).  CPU (oracle-backed tests) keeps 128.
PREFILL_CHUNK = int(os.environ.get("TRIFORCE_PREFILL_CHUNK", "4096"))
assert PREFILL_CHUNK % 128 == 0 and PREFILL_CHUNK > 0


def chunked_prefill(forward, input_ids):
    """Run ``forward(chunk) -> logits (1, rows, V)`` over the prompt; returns the logits of the reference's last chunk.
    A ``forward`` that takes ``last_rows`` is asked for only the rows that are returned — the trailing rows of the final
    chunk, one row of every other chunk — instead of a (4096, V) fp32 logits block (524 MB, ~1 TFLOP of lm_head GEMM)
    per chunk of which at most 128 rows were ever used."""
    import inspect
    n = input_ids.shape[-1]
    chunk = PREFILL_CHUNK if input_ids.is_cuda else 128
    keep = n - 128 * (math.ceil(n / 128) - 1)          # rows of the reference's last 128-token chunk
    try:
        trims = "last_rows" in inspect.signature(forward).parameters
    except (TypeError, ValueError):
        trims = False
    nchunks = math.ceil(n / chunk)
    for i in range(nchunks):
        ids = input_ids[:, i * chunk:(i + 1) * chunk]
        logits = forward(ids, last_rows=keep if i == nchunks - 1 else 1) if trims else forward(ids)
    return logits[:, -keep:]


class InferenceEngine:
    def __init__(self, model, cache, graph_cache, draft, draft_cache) -> None:
        self.model = model.eval()                  # target (7B)
        self.kv_cache = cache
        self.graph_cache = graph_cache
        self.draft = draft.eval()                  # 68M
        self.draft_cache = draft_cache

    @torch.inference_mode()
    def model_run(self, input_ids: torch.LongTensor, rebuild_retrieval=False):
        n = input_ids.shape[-1]
        if n > 64:                                 # chunked prefill (graph_infer.py:30-37)
            logits = chunked_prefill(lambda ids, last_rows=None: self.model.forward(
                ids, self.kv_cache, None, last_rows=last_rows).logits, input_ids)
        else:                                      # verification / q_len==1 retrieval build
            logits = self.model(input_ids=input_ids, kv_cache=self.kv_cache, graph_cache=self.graph_cache,
                                rebuild_retrieval=rebuild_retrieval).logits
        return logits

    @torch.inference_mode()
    def draft_run(self, input_ids: torch.LongTensor, gamma_offset: int = 0, probs=False, temperature=0.6, top_p=0.9):
        n = input_ids.shape[-1]
        if n > 64:                                 # draft prefill, 64 tokens per forward with eviction (:44-52)
            logits = self._draft_prefill(input_ids)
        else:
            # only the last row is used (graph_infer.py:57); rows are independent, so just that one is normalised —
            # inside the same native call when the draft runs through tf_draft_forward_68m
            out = self.draft.forward(input_ids, self.draft_cache, self.draft_cache, gamma_offset,
                                     probs=(temperature, top_p) if probs else None)
            return out.probs if probs else out.logits
        if probs:
            return norm_logits(logits[0, -1:], temperature=temperature, top_k=-1, top_p=top_p)[0]
        return logits

    def _draft_prefill(self, input_ids):
        """The 68M draft over the whole prompt, 64 tokens per forward with StreamingLLM eviction in front of each (reference
        graph_infer.py:44-52).  Once the window is full every full chunk is the SAME launch sequence — shift the window down
        by 64 rows, run 64 rows at slot start + recent - 64 — so on the device that step is captured once as a hipGraph and
        replayed (a 124 928-token prompt is 1 952 chunks of ~25 short launches each; eager they are host-bound).  The
        warm-up / capture passes run on a snapshot of the (1.6 MB) draft cache, which is restored afterwards; the ragged
        last chunk and the chunks that fill the window run eagerly.  TRIFORCE_DRAFT_PREFILL_GRAPH=0: all eager."""
        dc, n = self.draft_cache, input_ids.shape[-1]
        cap = dc.start_size + dc.recent_size
        chunks, full = math.ceil(n / 64), n // 64
        use_graph = (input_ids.is_cuda and full >= 16 and os.environ.get("TRIFORCE_DRAFT_PREFILL_GRAPH", "1") != "0"
                     and not torch.cuda.is_current_stream_capturing())
        logits, replayed = None, False
        for i in range(chunks):
            ids = input_ids[:, i * 64:(i + 1) * 64]
            if use_graph and ids.shape[-1] == 64 and dc.seq_len == cap:
                g = self._draft_prefill_graph(ids)
                if g is not None:
                    graph, ids_buf, logits = g
                    ids_buf.copy_(ids)
                    graph.replay()
                    dc.seq_len = cap               # what evict_prefill + the forward leave behind on the host side
                    replayed = True
                    continue
                use_graph = False
            dc.evict_prefill(64)
            logits, replayed = self.draft(input_ids=ids, kv_cache=dc, graph_cache=None).logits, False
        # the graph's logits are its static output buffer: the next prefill's replays overwrite it
        return logits.clone() if replayed else logits

    def _draft_prefill_graph(self, ids):
        """(graph, static 64-token input, static logits) of one steady-state draft-prefill step over the CURRENT draft
        cache and draft weights; captured at the first use, None if capture fails (the caller falls back to eager)."""
        dc = self.draft_cache
        key = (id(dc), dc.k.data_ptr(), id(self.draft))
        cached = getattr(self, "_dpf_graph", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        cap = dc.start_size + dc.recent_size
        snap = (dc.k.clone(), dc.v.clone(), dc.seq_len)
        ids_buf = ids.clone()

        def step():
            dc.seq_len = cap                       # host-side state at the entry of every steady step
            dc.evict_prefill(64)
            return self.draft(input_ids=ids_buf, kv_cache=dc, graph_cache=None).logits

        try:
            graph, logits = _capture(step, (), None, 1)
            out = (graph, ids_buf, logits)
        except Exception as ex:                    # capture unsupported for some op in this build: stay eager
            out = None
            print(f"[draft prefill] hipGraph capture of the steady-state step failed, running eagerly: "
                  f"{type(ex).__name__}: {ex}", flush=True)
        dc.k.copy_(snap[0])
        dc.v.copy_(snap[1])
        dc.seq_len = snap[2]
        self._dpf_graph = (key, out)
        return out

    @torch.inference_mode()
    def model_verify(self, input_ids: torch.LongTensor, position_ids: Optional[torch.LongTensor] = None, probs=False,
                     temperature=0.6, top_p=0.9):
        logits = self.model(input_ids=input_ids, kv_cache=self.kv_cache, graph_cache=self.graph_cache,
                            position_ids=position_ids, spec=True).logits
        if probs:
            return norm_logits(logits[0], temperature=temperature, top_k=-1, top_p=top_p)
        return logits

    def clear_kv(self):
        self.kv_cache.reset()
        self.graph_cache.reset()
        self.draft_cache.reset()


def _capture_error_mode():
    """hipStreamCaptureMode for our captures.  Once a torch.distributed NCCL (RCCL) process group exists, its watchdog
    thread polls hipEventQuery on outstanding collectives; under the default "global" mode such a call from another
    thread while this thread captures raises hipErrorStreamCaptureUnsupported INSIDE the watchdog, which terminates
    the process (observed on ROCm 7.2 / torch 2.10: tools/rccl_capture_probe.py).  "thread_local" confines the
    capture-safety check to the capturing thread — required for capturing anything, collectives included, in a TP
    process."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return "thread_local"
    except Exception:
        pass
    return "global"


def _graphable_cache():
    from ..models.cache import FlashSimpleCache
    return FlashSimpleCache


def _capture(fn, static_inputs, mempool, n_warmups):
    """Warm up on a side stream, then capture ``fn(*static_inputs)`` into one hipGraph.  Python's garbage collector is
    run first and held off while capturing: a collection that frees pinned host memory (decision-record mailboxes,
    offloaded caches of an engine that went out of scope) makes the host allocator record events on the capturing
    stream, which aborts the process."""
    gc.collect()
    gc_was_enabled = gc.isenabled()
    gc.disable()
    try:
        return _capture_nogc(fn, static_inputs, mempool, n_warmups)
    finally:
        if gc_was_enabled:
            gc.enable()


def _capture_nogc(fn, static_inputs, mempool, n_warmups):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(n_warmups):
            static_out = fn(*static_inputs)
        side.synchronize()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, pool=mempool, capture_error_mode=_capture_error_mode()):
        static_out = fn(*static_inputs)
    return graph, static_out


class _GraphedCall:
    """``fn(*static_inputs)`` captured once; a call copies the live arguments into the static input buffers, replays
    the hipGraph and returns a fresh copy of the static output (graph_infer.py:91-97,119-127 do exactly this for the
    two shapes below)."""

    def __init__(self, fn, static_inputs, mempool, n_warmups):
        self.inputs = static_inputs
        self.generation = 0                        # replays so far: a static output handed out at generation g is valid
        self.graph, self.output = _capture(fn, static_inputs, mempool, n_warmups)   # only while generation == g

    def __call__(self, *live, clone=True):
        for buf, value in zip(self.inputs, live):
            if not (value.data_ptr() == buf.data_ptr() and value.shape == buf.shape and value.stride() == buf.stride()):
                buf.copy_(value)                   # the caller may also write the static input buffer directly
        self.graph.replay()
        self.generation += 1
        # clone=False hands out the static output buffer itself: valid until THIS graph is replayed again (the decode
        # loops consume a draft / verify result before they ask for the next one; TriForceRunner.step asserts it through
        # ``generation``)
        return self.output.clone() if clone else self.output


class _InnerGraphs:
    """ONE hipGraph per inner position n of Middle_Spec (reference decoding.py:182-220): 68M draft step over n + 1 tokens ->
    draw d ~ q_d into the token buffer -> retrieval-cache verify of the gamma + 1 tokens -> accept test of d + follow-up
    draw, which also writes the follow-up token and the decision record (round 5).  The reference
    issues these as two graph replays with host-side sampling and three blocking reads in between; rounds 2-4 as two
    replays + two eager kernels + one record read.  Here an inner iteration is one launch and one record read: the draw /
    accept kernels sit inside the graph with their uniforms behind the stream's device cursor (ops.*_cur), the record is
    the LAST store of the graph's last kernel, behind write-through stores of the token id and the cursor."""

    def __init__(self, ge, gamma, rng, record):
        eng, kw = ge.engine, ge.sampling
        assert kw["probs"] and ge.tok_buf is not None and record.numel() >= 4
        self.key = _InnerGraphs.key_of(ge, gamma, rng, record)
        self.generation = 0
        self.graphs, self.p = [], []
        # Their OWN memory pool: captured into the engine's pool — after every other graph — their static outputs would be
        # carved from blocks the earlier captures had freed, i.e. from memory the target / draft graphs still use as scratch
        # at replay time, and the target verify runs between Middle_Spec and the outer accept test that reads these rows.
        self.pool = torch.cuda.graphs.graph_pool_handle()
        flat = ge.tok_buf.view(-1)
        ge.tok_buf.fill_(0)                                    # valid token ids for the warm-up passes
        # TRIFORCE_INNER_SPLIT=1 (A/B): the iteration as TWO graphs launched back to back — [draft step, draw] (14 nodes: its
        # packets reach the queue in a few us) and [retrieval verify, accept] (165 nodes, enqueued while the draft runs)
        self.split = os.environ.get("TRIFORCE_INNER_SPLIT", "0") == "1"
        for n in range(gamma):
            cell = {}

            def head(n=n, cell=cell):
                cell["q"] = eng.draft_run(input_ids=ge.tok_buf[:, :n + 1], gamma_offset=n, **kw)
                ops.sample_inverse_cdf_cur(cell["q"], rng.buf, rng.cursor, 0, flat[n + 1:n + 2])
                return cell["q"]

            def tail(n=n, cell=cell):
                p = eng.model_verify(input_ids=ge.tok_buf[:, :gamma + 1], position_ids=ge.pos_buf, **kw)
                ops.middle_accept_cur(p, cell["q"], flat, rng.buf, rng.cursor, n, gamma, record)
                return p

            def run(head=head, tail=tail):
                head()
                return tail()
            if self.split:
                g_head, q_static = _capture(head, (), self.pool, 2)
                cell["q"] = q_static
                g_tail, p = _capture(tail, (), self.pool, 2)
                self.graphs.append((g_head, g_tail))
            else:
                graph, p = _capture(run, (), self.pool, 2)
                self.graphs.append(graph)
            self.p.append(p)
        torch.cuda.synchronize()
        rng.device_cursor = False                              # the warm-up passes advanced the device copy

    @staticmethod
    def key_of(ge, gamma, rng, record):
        return (gamma, id(rng), rng.buf.data_ptr(), rng.cursor.data_ptr(), record.data_ptr(), ge.tok_buf.data_ptr(),
                id(ge.engine.kv_cache), id(ge.engine.graph_cache), id(ge.engine.draft_cache))

    def replay(self, n):
        g = self.graphs[n]
        if self.split:
            g[0].replay()
            g[1].replay()
        else:
            g.replay()
        self.generation += 1
        return self.p[n]


def draft_run_capture_graph(engine: InferenceEngine, gamma_offset: int = 0, mempool=None, n_warmups: int = 3, probs=False,
                            temperature=0.6, top_p=0.9, verbose=True, ids=None):
    """One 68M draft step over ``gamma_offset + 1`` tokens (reference graph_infer.py:74-97).  ``ids``: static input
    buffer to capture over (default: a fresh one)."""
    if verbose:
        print(f"[draft run] capturing graph for {gamma_offset} (probs={probs}, temp={temperature}, top_p={top_p})...")
    if ids is None:
        ids = torch.zeros((1, gamma_offset + 1), dtype=torch.long, device=engine.draft.device)
    return _GraphedCall(lambda t: engine.draft_run(input_ids=t, gamma_offset=gamma_offset, probs=probs,
                                                   temperature=temperature, top_p=top_p), (ids,), mempool, n_warmups)


def model_verify_capture_graph(engine: InferenceEngine, mempool=None, n_warmups: int = 3, gamma: int = 6, probs=False,
                               temperature=0.6, top_p=0.9, verbose=True, ids=None, pos=None):
    """The retrieval-cache verify over ``gamma + 1`` tokens at explicit positions (reference graph_infer.py:99-127)."""
    if verbose:
        print(f"[model verify] capturing graph for spec len {gamma} (probs={probs}, temp={temperature}, top_p={top_p})...")
    dev = engine.model.device
    if ids is None:
        ids = torch.zeros((1, gamma + 1), dtype=torch.long, device=dev)
    if pos is None:
        pos = torch.arange(gamma + 1, device=dev).unsqueeze(0)
    return _GraphedCall(lambda t, p: engine.model_verify(input_ids=t, position_ids=p, probs=probs, temperature=temperature,
                                                         top_p=top_p), (ids, pos), mempool, n_warmups)


class _TargetGraph:
    """The full-cache decode forward of ``q_len`` tokens (a target verify, or the q_len == 1 autoregressive step) as
    ONE hipGraph.  The reference keeps this forward eager because its key count changes every call
    (graph_infer.py:22-38); here the append slot and the key count are device scalars read by the kernels
    (tf_skinny_qkv_rope slot0_dev, tf_attn_decode sk_dev) and the launch is sized by the cache capacity, so one
    capture serves every cache length.  With ``probs`` the temperature / top-p normalisation is part of the graph."""

    def __init__(self, engine, q_len, mempool, n_warmups, probs, temperature, top_p, ids=None):
        dev = engine.model.device
        self.engine, self.q_len, self.probs = engine, q_len, probs
        self.cache = engine.kv_cache               # the graph reads and appends THIS cache's storage
        # ``ids``: capture over this (1, q_len) view instead of an own buffer — the engine passes the head of its shared token
        # buffer, which after the inner loop already holds [next, t_1 .. t_g2]: the verify then needs no token set-up at all
        self.ids = torch.zeros((1, q_len), dtype=torch.long, device=dev) if ids is None else ids
        assert tuple(self.ids.shape) == (1, q_len) and self.ids.is_contiguous()
        self.base = torch.arange(q_len, dtype=torch.long, device=dev)
        self.pos = torch.arange(q_len, dtype=torch.long, device=dev)
        self.slot = torch.zeros(1, dtype=torch.int32, device=dev)
        self.sk = torch.full((1,), q_len, dtype=torch.int32, device=dev)

        def run():
            logits = engine.model(input_ids=self.ids, kv_cache=engine.kv_cache, graph_cache=None,
                                  position_ids=self.pos.unsqueeze(0), dev_len=(self.slot, self.sk)).logits
            if probs:
                return logits, norm_logits(logits[0], temperature=temperature, top_k=-1, top_p=top_p)
            return logits, None

        self.graph, (self.logits, self.out_probs) = _capture(run, (), mempool, n_warmups)
        self._plan = None
        self.stale = True

    def __call__(self, input_ids):
        """input_ids: a (1, q_len) device tensor, or a python list of q_len ids (then they travel as kernel arguments)."""
        kvc = self.engine.kv_cache
        S = kvc.seq_len
        if S + self.q_len > kvc.max_budget:
            raise IndexError(f"FlashSimpleCache overflow: {S}+{self.q_len} > {kvc.max_budget}")
        self.stale = True                          # (scalars now encode THIS call's length: the owner's dev_len is void)
        if isinstance(input_ids, (list, tuple)):
            assert len(input_ids) == self.q_len <= 32
            if ops.HOST_PLANS and self.ids.is_cuda:
                if self._plan is None:                         # (validated once: ops.SetTokensPlan)
                    self._plan = ops.SetTokensPlan(self.ids.view(-1), self.pos, self.slot, self.sk)
                self._plan(input_ids, 0, pos0=S, sk_val=S + self.q_len)
            else:
                ops.set_tokens(self.ids, input_ids, 0, pos=self.pos, pos0=S, slot=self.slot, sk=self.sk, sk_val=S + self.q_len)
        elif self.ids.is_cuda and self.q_len <= 64 and os.environ.get("TRIFORCE_HOST_FAST", "1") != "0":   # (tf_set_tokens: <= 64 positions)
            self.ids.copy_(input_ids)
            ops.set_tokens(None, (), 0, pos=self.pos, pos0=S, slot=self.slot, sk=self.sk, sk_val=S + self.q_len)
        else:
            self.ids.copy_(input_ids)
            torch.add(self.base, S, out=self.pos)
            self.slot.fill_(S)
            self.sk.fill_(S + self.q_len)
        self.graph.replay()
        kvc.seq_len = S + self.q_len
        return self.logits, self.out_probs

    def replay_in_place(self):
        """The verify over the tokens ALREADY in ``ids`` with positions / slot / key count ALREADY on the device (left there by
        tf_accept_chain_step for the cache length the host also holds): the replay and the host-side length, nothing else."""
        kvc = self.engine.kv_cache
        S = kvc.seq_len
        if S + self.q_len > kvc.max_budget:
            raise IndexError(f"FlashSimpleCache overflow: {S}+{self.q_len} > {kvc.max_budget}")
        self.graph.replay()
        kvc.seq_len = S + self.q_len
        return self.out_probs


class GraphInferenceEngine:
    """The object the decode loops drive (reference graph_infer.py:129-194): ``inference`` /
    ``graph_draft_prefill`` run eagerly, ``graph_draft_inference`` / ``graph_verify`` replay the captured graphs."""

    def __init__(self, model, cache, graph_cache, draft, draft_cache) -> None:
        self.engine = InferenceEngine(model, cache, graph_cache, draft, draft_cache)
        self.callables = {}                        # gamma_offset -> draft step
        self.callable_model_verify = None
        self.target_graphs = {}                    # q_len -> _TargetGraph (full-cache forward, device-resident lengths)
        self.static_outputs = True                 # graph_draft_inference / graph_verify accept clone=False
        self.tok_buf = self.pos_buf = None         # shared static inputs of the draft / verify graphs (graphs only)
        self._inner = {}                           # inner-iteration graphs per (rng, record): see inner_graphs()
        self.dev_len = None
        self.mempool = None
        self.sampling = dict(probs=False, temperature=0.6, top_p=0.9)

    @torch.inference_mode()
    def initialize_cuda_graph(self, gamma=6, probs=False, temperature=0.6, top_p=0.9, verbose=True,
                              capture_target=True):
        """gamma + 3 draft graphs (one per gamma_offset) and one retrieval-verify graph, sharing one memory pool."""
        gc.collect()
        self.mempool = torch.cuda.graphs.graph_pool_handle()
        self.sampling = dict(probs=probs, temperature=temperature, top_p=top_p)
        common = dict(engine=self.engine, mempool=self.mempool, n_warmups=3, verbose=verbose, **self.sampling)
        # ONE token buffer is the static input of every draft graph (its first gamma_offset + 1 entries) and of the
        # retrieval-verify graph (its first gamma + 1): the decode loop writes drafted tokens straight into it (the
        # sampling / accept kernels do) and no per-replay input copy is left.  Same for the verify positions.
        dev = self.engine.model.device
        self._inner = {}
        self.tok_buf = torch.zeros((1, gamma + 3), dtype=torch.long, device=dev)
        self.pos_buf = torch.arange(gamma + 1, device=dev).unsqueeze(0).clone()
        self.callables = {off: draft_run_capture_graph(gamma_offset=off, ids=self.tok_buf[:, :off + 1], **common)
                          for off in range(gamma + 3)}
        self.callable_model_verify = model_verify_capture_graph(gamma=gamma, ids=self.tok_buf[:, :gamma + 1],
                                                                pos=self.pos_buf, **common)
        self.target_graphs = {}
        if capture_target and os.environ.get("TRIFORCE_TARGET_GRAPH", "1") != "0" \
                and isinstance(self.engine.kv_cache, _graphable_cache()):
            # target verify (gamma+1 / gamma+2 tokens: Middle_Spec ends at n = gamma or gamma + 1) and the AR step
            for q_len in sorted({1, gamma + 1, gamma + 2}):
                # the two verify lengths read their tokens from the head of the shared token buffer (see _TargetGraph)
                self.target_graphs[q_len] = _TargetGraph(self.engine, q_len, self.mempool, 3, probs and q_len > 1,
                                                         temperature, top_p, ids=self.tok_buf[:, :q_len] if q_len > 1 else None)
        self.dev_len = None                        # cache length the verify graphs' device scalars encode (None: unknown)
        self.engine.clear_kv()

    def initialize_eager(self, gamma=6, probs=True, temperature=0.6, top_p=0.9):
        """Same surface without graph capture (debugging / parity bisecting)."""
        self.sampling = dict(probs=probs, temperature=temperature, top_p=top_p)
        eng, kw = self.engine, self.sampling
        self.callables = {off: (lambda ids, off=off: eng.draft_run(input_ids=ids, gamma_offset=off, **kw))
                          for off in range(gamma + 3)}
        self.callable_model_verify = lambda ids, pos: eng.model_verify(input_ids=ids, position_ids=pos, **kw)

    def clear_kv(self):
        self.engine.clear_kv()

    # -- the surface used by utils/decoding.py ---------------------------------------------------------------
    @torch.inference_mode()
    def inference(self, input_ids: torch.LongTensor, rebuild_retrieval=False, eager=False):
        tg = None if (eager or rebuild_retrieval) else self._target_graph(input_ids)
        if tg is not None:
            return tg(input_ids)[0].clone()
        return self.engine.model_run(input_ids=input_ids, rebuild_retrieval=rebuild_retrieval)

    def _target_graph(self, input_ids):
        """The captured full-cache forward for a verify block, if there is one.  q_len == 1 through ``inference`` is the
        call that (re)builds the retrieval cache (reference modeling_llama.py:226-238) and stays eager; the plain
        autoregressive step has its own entry, ``decode_step``."""
        if input_ids.shape[0] != 1 or input_ids.shape[-1] == 1:
            return None
        tg = self.target_graphs.get(input_ids.shape[-1])
        return tg if (tg is not None and tg.cache is self.engine.kv_cache) else None   # cache swapped after capture: eager

    @torch.inference_mode()
    def verify_probs(self, input_ids, temperature, top_p, rebuild_retrieval=False, eager=False):
        """Target verify -> (rows, V) probabilities after temperature / top-p: one graph replay when captured with
        these settings, else the eager forward + norm_logits.  The result is valid until the next target forward."""
        tg = None if (eager or rebuild_retrieval) else self._target_graph(input_ids)
        if tg is not None and tg.probs and (temperature, top_p) == (self.sampling["temperature"], self.sampling["top_p"]):
            return tg(input_ids)[1]
        logits = self.inference(input_ids, rebuild_retrieval=rebuild_retrieval, eager=eager)
        return norm_logits(logits[0], temperature=temperature, top_k=-1, top_p=top_p)

    @torch.inference_mode()
    def verify_probs_ids(self, ids, temperature, top_p):
        """``verify_probs`` for a python list of token ids when the captured target graph of that length exists with these
        sampling settings: (probabilities, the graph's (1, q_len) device token row), else None (caller: tensor path).
        One launch sets ids, positions and lengths (tf_set_tokens) in front of the replay."""
        tg = self.target_graphs.get(len(ids))
        if tg is None or tg.cache is not self.engine.kv_cache or not tg.probs or not tg.ids.is_cuda \
                or (temperature, top_p) != (self.sampling["temperature"], self.sampling["top_p"]):
            return None
        return tg(list(ids))[1], tg.ids

    def verify_sets(self, gamma):
        """(pos, slot, sk, q_len) of the captured verify lengths gamma + 1 / gamma + 2 — what tf_accept_chain_step keeps current —
        or None when one of them is missing, bound to another cache, or not fed from the shared token buffer."""
        out = []
        for q_len in (gamma + 1, gamma + 2):
            tg = self.target_graphs.get(q_len)
            if tg is None or tg.cache is not self.engine.kv_cache or not tg.probs or self.tok_buf is None \
                    or tg.ids.data_ptr() != self.tok_buf.data_ptr():
                return None
            out.append((tg.pos, tg.slot, tg.sk, q_len))
        return out

    def sync_verify_lengths(self, gamma):
        """Bring the verify graphs' device scalars to the cache's current length (one launch per graph; only when they are stale:
        first step of a prompt, after an eager / rebuild / autoregressive forward)."""
        S = self.engine.kv_cache.seq_len
        for q_len in (gamma + 1, gamma + 2):
            tg = self.target_graphs[q_len]
            ops.set_tokens(None, (), 0, pos=tg.pos, pos0=S, slot=tg.slot, sk=tg.sk, sk_val=S + q_len)
            tg.stale = False
        self.dev_len = S

    def verify_lengths_current(self, gamma):
        return self.dev_len == self.engine.kv_cache.seq_len and not any(self.target_graphs[q].stale for q in (gamma + 1, gamma + 2))

    @torch.inference_mode()
    def decode_step(self, input_ids):
        """One autoregressive step (q_len == 1, no retrieval build) over the full cache: the captured graph when there
        is one, else the eager forward the reference runs (decoding.py:28)."""
        tg = self.target_graphs.get(1)
        if tg is not None and tg.cache is self.engine.kv_cache and input_ids.shape == (1, 1):
            return tg(input_ids)[0]
        return self.engine.model(input_ids=input_ids, kv_cache=self.engine.kv_cache, graph_cache=None).logits

    @torch.inference_mode()
    def graph_draft_prefill(self, input_ids: torch.LongTensor):
        return self.engine.draft_run(input_ids=input_ids)

    @torch.inference_mode()
    def graph_draft_inference(self, input_ids: torch.LongTensor, gamma_offset: int = 0, clone=True):
        fn = self.callables[gamma_offset]
        return fn(input_ids, clone=clone) if isinstance(fn, _GraphedCall) else fn(input_ids)

    def replay_draft(self, gamma_offset):
        """graph_draft_inference(tok_buf[:, :gamma_offset + 1], gamma_offset, clone=False) for a caller that has written
        the tokens into ``tok_buf`` itself: the replay and nothing else."""
        fn = self.callables[gamma_offset]
        if not isinstance(fn, _GraphedCall):
            return fn(self.tok_buf[:, :gamma_offset + 1])
        fn.graph.replay()
        fn.generation += 1
        return fn.output

    @torch.inference_mode()
    def graph_verify(self, input_ids: torch.LongTensor, position_ids: torch.LongTensor, clone=True):
        fn = self.callable_model_verify
        return fn(input_ids, position_ids, clone=clone) if isinstance(fn, _GraphedCall) else fn(input_ids, position_ids)

    def verify_generation(self):
        """Replay count of the retrieval-verify graph(s) (None when not captured): the lifetime token of the static
        probability rows ``graph_verify(..., clone=False)`` / an inner-iteration graph hands out."""
        fn = self.callable_model_verify
        if not isinstance(fn, _GraphedCall):
            return None
        return fn.generation + sum(g.generation for g in self._inner.values())

    @torch.inference_mode()
    def inner_graphs(self, gamma, rng, record, capture=True):
        """The per-position inner-iteration graphs (``_InnerGraphs``) over this engine's token buffer, ``rng``'s buffer /
        cursor and the pinned decision ``record`` — captured at the first request (outside any timed region: the decode
        runner asks for them when it is built), cached per (rng, record); None when the engine cannot provide them
        (eager engine, logits instead of probabilities, CPU, TRIFORCE_INNER_GRAPH=0)."""
        if os.environ.get("TRIFORCE_INNER_GRAPH", "1") == "0" or self.tok_buf is None or not self.tok_buf.is_cuda \
                or not self.sampling.get("probs") or not isinstance(self.callable_model_verify, _GraphedCall) \
                or self.tok_buf.shape[1] < gamma + 3 or record.numel() < 4 or record.is_cuda \
                or torch.cuda.is_current_stream_capturing():
            return None
        key = _InnerGraphs.key_of(self, gamma, rng, record)
        got = self._inner.get(key)
        if got is None and not capture:                        # (Middle_Spec only uses what its runner captured)
            return None
        if got is None:
            if len(self._inner) >= 2:                          # a runner per prompt / per seed: keep the two most recent sets
                self._inner.pop(next(iter(self._inner)))
            got = self._inner[key] = _InnerGraphs(self, gamma, rng, record)
        return got

    def init_graph_cache(self):
        self.engine.graph_cache.init_graph_cache(kv_cache=self.engine.kv_cache)

    def update_graph_cache(self):
        self.engine.graph_cache.update_graph_cache(kv_cache=self.engine.kv_cache)
