"""Decode algorithms of the TriForce path — same entry points as the reference's utils/decoding.py:
Autoregressive (:14-37), TriForce (:41-160), Middle_Spec (:163-223) (+ the TP variants in
decoding_dist.py).  Same algorithm, same quirks, different execution:

  * every accept/reject decision, residual resample and bonus sample runs on the device
    (tf_accept_chain / tf_middle_accept / tf_sample_inverse_cdf); the host reads back ONE small int64
    record per inner step and ONE per outer step instead of >= gamma+2 blocking ``.item()`` /
    ``if r < ...`` syncs (decoding.py:97-121,190-193);
  * randomness comes from an explicit uniform stream (utils.sampling.UniformSource) consumed in
    the reference's decision order, so a run is reproducible and comparable across devices;
  * timing brackets the decode loop with device synchronisation (the reference omits it on-chip).
"""
import time

import numpy as np
import torch

from .. import ops
from .misc import log_csv, spec_stream
from .sampling import UniformSource, norm_logits, sample

PAD_TOKEN = 100            # filler id of verify_tokens / pass_tokens (decoding.py:94,177)
# 0: round 3's host path between a record read and the next launch (ids through pinned staging, separate position / length
#    launches, input checks on every draft replay) — kept for same-box A/B of DESIGN 13.6; 1 (default): ids as kernel arguments
#    TRIFORCE_HOST_FAST_MASK selects pieces: 1 ids / positions of Middle_Spec, 2 check-free draft replay, 4 target verify,
#    8 catch-up draft input, 16 the remaining host -> device token lists (eager / rebuild steps) without an H2D copy
_HF = __import__("os").environ
HOST_FAST_MASK = 0 if _HF.get("TRIFORCE_HOST_FAST", "1") == "0" else int(_HF.get("TRIFORCE_HOST_FAST_MASK", "31"))
HOST_FAST = HOST_FAST_MASK != 0


def _sync(device):
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)


def _eos(tokenizer):
    e = getattr(tokenizer, "eos_token_id", None)
    return -1 if e is None else int(e)


@torch.inference_mode()
def Autoregressive(tokenizer, graph_engine, input_ids, max_len=256, top_k=-1, top_p=0.9, temperature=0.6, verbose=False,
                   rng=None, return_tokens=False):
    eng = graph_engine.engine
    device = eng.model.device
    rng = rng or UniformSource(device)
    eng.kv_cache.reset()
    logits = graph_engine.inference(input_ids=input_ids)
    next_token = sample(norm_logits(logits[:, -1, :], temperature=temperature, top_k=top_k, top_p=top_p), rng=rng)
    toks = [next_token]
    n = 0
    _sync(device)
    time1 = time.time()
    step = getattr(graph_engine, "decode_step", None) or \
        (lambda t: eng.model(input_ids=t, kv_cache=eng.kv_cache, graph_cache=None).logits)
    while n < max_len:      # no host sync inside the loop: the sampled token never leaves the device
        logits = step(next_token)
        next_token = sample(norm_logits(logits[:, -1, :], temperature=temperature, top_k=top_k, top_p=top_p), rng=rng)
        toks.append(next_token)
        n += 1
    _sync(device)
    time2 = time.time()
    if verbose or return_tokens:
        ids = torch.cat(toks, dim=1)[0].tolist()
        if verbose:
            for t in ids:
                spec_stream(t, tokenizer, "cyan")
        if return_tokens:
            return n / (time2 - time1), ids
    return n / (time2 - time1)


# The host waits for a decision record by polling pinned memory.  A poll loop that never yields can sit on the core a HIP
# runtime thread needs (launch hand-off, completion signals): measured on 8-vCPU boxes, the hop between an accept kernel
# and the next launch was 12 us in one process and 61 us in another for the same code (profiles/r04_host_path_ab.txt).
# TRIFORCE_POLL_YIELD=1: sched_yield() per poll.
_POLL_YIELD = __import__("os").environ.get("TRIFORCE_POLL_YIELD", "0") == "1"
# The replay that follows a record read spends 36-69 us INSIDE hipGraphLaunch in the loop (tools/hop_trace.py) against ~9 us
# from an idle, synchronised queue (tools/draft_launch_latency.py): the runtime retires the completed commands of the
# 165-node verify graph at its next API call.  TRIFORCE_POLL_QUERY=1: hipStreamQuery every 16th poll, so that the
# retirement happens while the host waits anyway.
_POLL_QUERY = __import__("os").environ.get("TRIFORCE_POLL_QUERY", "0") == "1"
# The record becomes visible a moment BEFORE its kernel completes; a hipGraphLaunch into a stream whose last command is still
# in flight takes 35-56 us on the host here, into an idle stream ~9 us in all.  TRIFORCE_SYNC_AFTER_RECORD=1: wait for the
# stream to drain (the kernel is ending anyway) before the replay.
_SYNC_AFTER_RECORD = __import__("os").environ.get("TRIFORCE_SYNC_AFTER_RECORD", "0") == "1"
_POLL_QUERY_MASK = int(__import__("os").environ.get("TRIFORCE_POLL_QUERY_EVERY", "16")) - 1
if (_POLL_QUERY_MASK + 1) & _POLL_QUERY_MASK or _POLL_QUERY_MASK < 0:
    raise ValueError("TRIFORCE_POLL_QUERY_EVERY must be a power of two (it is used as a mask)")
_sched_yield = getattr(__import__("os"), "sched_yield", lambda: None)


# TRIFORCE_HOP_TRACE=1: (ns from "record seen" to "replay called", ns inside the replay call) per inner hop, for tools
_HOP_TRACE = [] if __import__("os").environ.get("TRIFORCE_HOP_TRACE", "0") == "1" else None


class _Record:
    """A small int64 decision record written by a kernel and read by the host once per (inner / outer) step.
    mailbox=True: the record lives in PINNED HOST memory that the kernel writes directly (unified addressing); the host
    arms it with a sentinel before the launch and polls — no device-to-host copy, no stream synchronisation (~20 us of
    idle GPU per read otherwise, 5 reads per step).  Else: a device tensor read with .tolist() (what TP needs: the
    record is broadcast over RCCL first)."""

    SENTINEL = -(1 << 62)

    def __init__(self, device, n, mailbox):
        self.n = n
        self.mailbox = bool(mailbox) and torch.device(device).type == "cuda"
        if self.mailbox:
            self.tensor = torch.zeros(n, dtype=torch.int64).pin_memory()
            self._np = self.tensor.numpy()
        else:
            self.tensor = torch.zeros(n, dtype=torch.int64, device=device)

    def arm(self, k):
        if self.mailbox:
            self._np[:k] = self.SENTINEL

    def read(self, k):
        if not self.mailbox:
            return self.tensor[:k].tolist()
        view, s, spins = self._np[:k], self.SENTINEL, 0
        query = torch.cuda.current_stream().query if _POLL_QUERY else None
        while (view == s).any():
            spins += 1
            if query is not None and (spins & _POLL_QUERY_MASK) == 0:
                query()                # lets the HIP runtime retire finished commands NOW (see _POLL_QUERY)
            if _POLL_YIELD:
                _sched_yield()         # let the HIP runtime's own threads run if they share this core (see _POLL_YIELD)
            if spins > 50_000_000:
                raise RuntimeError("decision record never arrived (kernel failed?)")
        return view.tolist()


# One hipGraph per inner iteration (draft step, draw, retrieval verify, accept test: utils/graph_infer._InnerGraphs) instead
# of two replays + two eager kernels.  TRIFORCE_INNER_GRAPH=0: rounds 2-4's four launches.
# (Round 5 also built the two-stream form the round-4 trace suggested — every chain of launches between two record reads on
#  alternating streams, ordered by the record alone — and measured it on the same box against this form: +150..200 us per step
#  (profiles/r05_inner_loop_ab.jsonl: a launch into the OTHER hardware queue starts later than one behind the still-completing
#  accept kernel); it is also unsound without an explicit acquire at the head of every chain — the runtime knows nothing of an
#  ordering that goes through host memory.  Removed)
INNER_GRAPH = __import__("os").environ.get("TRIFORCE_INNER_GRAPH", "1") != "0"
# With the inner graphs: the outer accept kernel also writes the catch-up draft's pass tokens and the next target verify's
# positions / slot / key count on the device (tf_accept_chain_step), and the target verify reads its tokens from the shared token
# buffer — no set-up launch behind either record.  TRIFORCE_STEP_ON_DEVICE=0: the host sets them up (tf_set_tokens).
STEP_ON_DEVICE = __import__("os").environ.get("TRIFORCE_STEP_ON_DEVICE", "1") != "0"


_MAILBOX_OK = {}


def _mailbox_supported(device):
    """One-off probe per device: a kernel writes a pinned host record and the host sees it without any synchronisation
    (coherent pinned memory).  TRIFORCE_MAILBOX=0 forces the copy-back path."""
    import os
    key = str(torch.device(device))
    if key not in _MAILBOX_OK:
        ok = False
        if torch.device(device).type == "cuda" and os.environ.get("TRIFORCE_MAILBOX", "1") != "0":
            try:
                rec = _Record(device, 4, True)
                probs = torch.zeros(64, dtype=torch.float32, device=device)
                probs[5] = 1.0
                rec.arm(1)
                ops.sample_inverse_cdf(probs, torch.full((1,), 0.5, device=device), rec.tensor[:1])
                t0 = time.time()
                while int(rec._np[0]) == rec.SENTINEL and time.time() - t0 < 0.25:
                    pass
                ok = int(rec._np[0]) == 5
                torch.cuda.synchronize(device)
            except Exception:
                ok = False
        _MAILBOX_OK[key] = ok
    return _MAILBOX_OK[key]


class _SpecBuffers:
    """Per-engine device scratch reused across iterations (allocated once)."""

    def __init__(self, device, gamma, vocab, graph_engine=None, mailbox=False):
        tok_buf = getattr(graph_engine, "tok_buf", None)
        self.shared_inputs = tok_buf is not None and tok_buf.shape[1] >= gamma + 1
        if self.shared_inputs:
            # the engine's graphs read their tokens / positions from these very buffers: no input copies per replay
            self.verify_tokens = tok_buf[:, :gamma + 1]
            self.positions = graph_engine.pos_buf
        else:
            self.verify_tokens = torch.full((1, gamma + 1), PAD_TOKEN, dtype=torch.long, device=device)
            self.positions = torch.zeros((1, gamma + 1), dtype=torch.long, device=device)
        self.start_plan = self.pass_plan = None      # ops.SetTokensPlan of Middle_Spec's first launch / of the pass tokens
        self.inner_for = None                        # (inner-iteration graphs, the uniform stream they were captured for)
        self.spec_rows = torch.empty(gamma + 2, vocab, dtype=torch.float32, device=device)
        self.rows_generation = None                # lifetime token of static (un-copied) retrieval-verify rows
        mailbox = bool(mailbox) and _mailbox_supported(device)
        self.mid_out = _Record(device, 4, mailbox)
        self.chain_out = _Record(device, 4, mailbox)
        self.pos_base = torch.arange(gamma + 1, dtype=torch.long, device=device).unsqueeze(0)
        # host -> device token lists go through one pinned staging row (a pageable source makes the copy synchronous)
        cuda = torch.device(device).type == "cuda"
        self.stage = torch.zeros(2, gamma + 4, dtype=torch.long, pin_memory=cuda)
        self.dev_tokens = torch.zeros(2, gamma + 4, dtype=torch.long, device=device)
        self._flip = 0

    def to_device(self, ids):
        """(1, len(ids)) int64 device tensor holding the python list ``ids``.  Two (staging, device) row pairs
        alternate: the row handed out by the previous call stays intact while this one is filled, and a staging row is
        rewritten only after a host sync (every decode step reads a decision record between two calls) has retired the
        copy that read it."""
        n = len(ids)
        self._flip ^= 1
        row = self.dev_tokens[self._flip, :n]
        if (HOST_FAST_MASK & 16) and row.is_cuda and n <= 32:
            ops.set_tokens(row, ids, PAD_TOKEN)                # ids as kernel arguments: no staging copy, no H2D copy
            return row.unsqueeze(0)
        stage = self.stage[self._flip, :n]
        stage.copy_(torch.tensor(ids, dtype=torch.long))
        row.copy_(stage, non_blocking=True)
        return row.unsqueeze(0)


def _buffers(graph_engine, gamma, vocab, device, mailbox=False):
    b = getattr(graph_engine, "_tf_spec_buffers", None)
    stale = b is not None and getattr(graph_engine, "tok_buf", None) is not None \
        and b.verify_tokens.data_ptr() != graph_engine.tok_buf.data_ptr()        # graphs were re-captured
    if b is None or stale or b.verify_tokens.shape[1] != gamma + 1 or b.spec_rows.shape[1] != vocab \
            or b.mid_out.mailbox != (bool(mailbox) and _mailbox_supported(device)):
        b = _SpecBuffers(device, gamma, vocab, graph_engine, mailbox)
        graph_engine._tf_spec_buffers = b
    return b


@torch.inference_mode()
def Middle_Spec(next_token, graph_engine, gamma, verbose, tokenizer, rng=None, buffers=None, sync_record=None,
                health=None):
    """Inner loop: the 68M drafts one token at a time for the retrieval-cache model (decoding.py:163-223).
    Returns (ids [next, t1..t_g2], rows = device (g2, V) view of the retrieval-model prob rows, acceptance)."""
    eng = graph_engine.engine
    device = eng.model.device
    rng = rng or UniformSource(device)
    if buffers is None:
        buffers = _buffers(graph_engine, gamma, eng.model.config.vocab_size, device, mailbox=sync_record is None)
    inner = None
    if INNER_GRAPH and sync_record is None and buffers.mid_out.mailbox and buffers.shared_inputs:
        cached = getattr(buffers, "inner_for", None)           # (graphs, rng) the runner captured for exactly this stream
        if cached is not None and cached[1] is rng and cached[0].key[0] == gamma:
            inner = cached[0]
        elif hasattr(graph_engine, "inner_graphs"):
            inner = graph_engine.inner_graphs(gamma, rng, buffers.mid_out.tensor, capture=False)
    S = eng.kv_cache.seq_len
    n = accepted = drafted = 0
    ids = [int(next_token)]
    vt = buffers.verify_tokens
    # [next, PAD...] and the gamma + 1 positions S, S + 1, ... in one launch (the token id travels as a kernel argument)
    if (HOST_FAST_MASK & 1) and vt.numel() <= 32 and buffers.positions.numel() <= 64:      # (tf_set_tokens' limits)
        position_ids = buffers.positions
        if ops.HOST_PLANS and vt.is_cuda:
            if buffers.start_plan is None:
                buffers.start_plan = ops.SetTokensPlan(vt.view(-1), position_ids)
            buffers.start_plan(ids, PAD_TOKEN, pos0=S)
        else:
            ops.set_tokens(vt, ids, PAD_TOKEN, pos=position_ids, pos0=S)
    else:
        vt.fill_(PAD_TOKEN)
        vt[0, 0] = ids[0]
        position_ids = torch.add(buffers.pos_base, S, out=buffers.positions)
    # the engines' graph replays can hand out their static output buffers (valid until the same graph replays again):
    # q_d and p are consumed by tf_middle_accept / the spec_rows copies before the next iteration asks for new ones
    noclone = dict(clone=False) if getattr(graph_engine, "static_outputs", False) else {}
    # the draft / verify graphs read vt itself: replay without looking at the inputs (this call sits between the host's
    # read of the previous accept record and the next launch — the one place where host time is GPU idle time)
    replay_draft = getattr(graph_engine, "replay_draft", None) if ((HOST_FAST_MASK & 2) and noclone and buffers.shared_inputs) else None
    flat = vt.view(-1)
    while inner is not None and n < gamma:
        # one launch, one read: [draft step n, draw, retrieval verify, accept test + follow-up draw] is ONE hipGraph whose
        # kernels take their three uniforms from the stream's device cursor and advance it
        rng.cursor_tensor(3)
        rec = buffers.mid_out
        rec.arm(4)
        if _HOP_TRACE is not None and drafted:
            _t_before = time.perf_counter_ns()
        p = inner.replay(n)
        if _HOP_TRACE is not None and drafted:                                # host pieces of the inner hop (tools/hop_trace.py)
            _HOP_TRACE.append((_t_before - _t_seen, time.perf_counter_ns() - _t_before))
        acc, b, d, at = rec.read(4)                                           # the one host read of this step
        if _HOP_TRACE is not None:
            _t_seen = time.perf_counter_ns()
        rng.advanced_on_device(3, at)
        drafted += 1
        if acc:                                                               # decoding.py:193-209
            ids += [d, b]
            accepted += 1
            n += 2
        else:                                                                 # decoding.py:211-220
            ids.append(b)
            n += 1
        if verbose:
            for t, colour in (((d, "green"), (b, "blue")) if acc else ((b, "red"),)):
                spec_stream(t, tokenizer, colour)
    if inner is not None:
        # row i of the LAST replay's static output is the retrieval model's distribution after tokens 0..i (see below)
        buffers.rows_generation = graph_engine.verify_generation()
        return ids, p[:len(ids) - 1], accepted / drafted
    while n < gamma:
        if _HOP_TRACE is not None and drafted:
            _t_before = time.perf_counter_ns()
        if replay_draft is not None:
            q_d = replay_draft(n)
            if _HOP_TRACE is not None and drafted:            # host pieces of the inner hop (TRIFORCE_HOP_TRACE=1, tools only)
                _HOP_TRACE.append((_t_before - _t_seen, time.perf_counter_ns() - _t_before))
        else:
            q_d = graph_engine.graph_draft_inference(input_ids=vt[:, :n + 1], gamma_offset=n, **noclone)
        u = rng.take(3)
        ops.sample_inverse_cdf(q_d, u[0:1], flat[n + 1:n + 2])               # d ~ q_d, written into verify_tokens
        if sync_record is not None:           # TP: rank 0's draft token is everyone's BEFORE the all-reduced verify runs
            sync_record(flat[n + 1:n + 2])    # on it (the reference's sample_dist, decoding.py:230-239,452)
        p = graph_engine.graph_verify(input_ids=vt, position_ids=position_ids, **noclone)
        rec = buffers.mid_out
        rec.arm(3)
        ops.middle_accept(p, q_d, flat, u[1:3], n, gamma, rec.tensor)         # accept test + follow-up sample
        if sync_record is not None:                                           # TP: rank 0's decision wins
            sync_record(rec.tensor)
            if n + 1 < flat.numel() and not _single_rank():                   # (one rank: the record IS the kernel's own)
                if flat.is_cuda:
                    ops.mid_record_tokens(rec.tensor, flat, n)               # one launch (the torch form below: six)
                else:
                    flat[n + 1:n + 3].copy_(_mid_tokens(rec.tensor, n, gamma, flat))
        acc, b, d = rec.read(3)                                               # the one host read of this step
        if _HOP_TRACE is not None:
            _t_seen = time.perf_counter_ns()                # (the drain below counts as part of the hop)
        if _SYNC_AFTER_RECORD:
            torch.cuda.current_stream().synchronize()       # the accept kernel is ending: launch from an IDLE stream (see there)
        if health is not None:                # TP: a timed-out exchange NaN-filled p — stop before its tokens are used
            health()
        rng.advance(3)
        drafted += 1
        g = len(ids) - 1                                                      # == n: verify_tokens holds exactly ids
        if not noclone:
            buffers.spec_rows[g].copy_(p[n])                                  # decoding.py:193-220: q row(s) of this step
            if acc:
                buffers.spec_rows[g + 1].copy_(p[n + 1])
        if acc:                                                               # decoding.py:193-209
            ids += [d, b]
            accepted += 1
            n += 2
            if verbose:
                spec_stream(d, tokenizer, "green")
                spec_stream(b, tokenizer, "blue")
        else:                                                                 # decoding.py:211-220
            ids.append(b)
            n += 1
            if verbose:
                spec_stream(b, tokenizer, "red")
    if noclone:
        # Static verify output: row i of the LAST replay is the retrieval model's distribution after tokens 0..i —
        # the very row an earlier replay produced when position i was decided (same graph, same tokens <= i, same
        # cache; the kernels are deterministic), so the rows need not be copied out step by step.  The rows stay valid
        # only until the verify graph replays again: the caller checks ``rows_generation`` before consuming them.
        gen = getattr(graph_engine, "verify_generation", None)
        buffers.rows_generation = gen() if gen is not None else None
        return ids, p[:len(ids) - 1], accepted / drafted
    buffers.rows_generation = None
    return ids, buffers.spec_rows[:len(ids) - 1], accepted / drafted


def _single_rank():
    import torch.distributed as _d
    return not (_d.is_available() and _d.is_initialized() and _d.get_world_size() > 1)


def _mid_tokens(rec, n, gamma, flat):
    """verify_tokens[n+1 : n+3] implied by a (possibly broadcast) middle_accept record (acc, b, d)."""
    acc, b, d = rec[0], rec[1], rec[2]
    cur = flat[n + 1:n + 3].clone()
    first = torch.where(acc > 0, d, b)                 # accepted: d stays at n+1, b goes to n+2; rejected: b at n+1
    cur[0] = first
    if cur.numel() > 1:
        cur[1] = torch.where(acc > 0, b, cur[1])
    return cur


class TriForceRunner:
    """The TriForce outer loop (decoding.py:41-160) as prefill() + step(): ``TriForce`` below drives it to
    ``max_len`` tokens, bench.py drives it for an exact number of steps."""

    def __init__(self, tokenizer, graph_engine, gamma, top_k=-1, top_p=0.9, temperature=0.6, verbose=False, rng=None,
                 inclusive_accept=False, sync_record=None, rebuild_every=0):
        # rebuild_every: N > 0 re-selects the retrieval cache's prefill chunks during every N-th target verify
        #                (SURVEY 8f row 4); 0 = the reference's behaviour, one build per prompt
        self.rebuild_every, self.rebuilds = int(rebuild_every), 0
        # eager_every: N > 0 runs every N-th target verify eagerly instead of replaying its hipGraph, so that
        #              bench.py can bracket individual attention launches with HIP events inside the timed region
        self.eager_every = 0
        # inclusive_accept: marks the TP outer loop — it tests ``r <=`` (decoding.py:347) where on-chip tests ``r <``
        #              (:99), and it ends at an eos that closes the accept scan (:382-383) where on-chip keeps going
        # sync_record: optional callable applied to each device decision record before it is read (TP: broadcast
        #              from rank 0, the role of sample_dist / the r broadcast, decoding.py:230-239,345-346)
        self.inclusive_accept, self.sync_record = inclusive_accept, sync_record
        # health: optional callable run once per outer step right after the step's host read (TP: raises when the
        #              one-shot all-reduce has timed out — its outputs are NaN-filled from then on, csrc/allreduce.hip)
        self.health = None
        self.tokenizer, self.ge, self.eng = tokenizer, graph_engine, graph_engine.engine
        self.gamma, self.top_k, self.top_p, self.temperature, self.verbose = gamma, top_k, top_p, temperature, verbose
        self.device = self.eng.model.device
        self.rng = rng or UniformSource(self.device)
        self.eos = _eos(tokenizer)
        self.bufs = _buffers(graph_engine, gamma, self.eng.model.config.vocab_size, self.device,
                             mailbox=sync_record is None)
        self.inner = None
        if INNER_GRAPH and sync_record is None and self.bufs.mid_out.mailbox and self.bufs.shared_inputs \
                and hasattr(graph_engine, "inner_graphs"):
            self.inner = graph_engine.inner_graphs(gamma, self.rng, self.bufs.mid_out.tensor)   # captured here, not in a step
            self.bufs.inner_for = (self.inner, self.rng) if self.inner is not None else None
        self.resample_count = self.accepted_count = self.target_sample_count = self.draft_count = 0
        self.n = 0
        self.inner_iters = 0          # Middle_Spec iterations = 68M draft calls = retrieval-verify replays
        self.emitted, self.counts, self.acc_rate_middle_list = [], [], []
        self.next_token, self.last_reason = None, None

    @torch.inference_mode()
    def prefill(self, input_ids):
        eng, ge = self.eng, self.ge
        eng.kv_cache.reset()
        eng.graph_cache.reset()
        eng.draft_cache.reset()
        ge.inference(input_ids=input_ids[:, :-1])
        logits = ge.inference(input_ids=input_ids[:, -1:])         # q_len==1 -> the retrieval cache is built here
        ge.graph_draft_prefill(input_ids=input_ids)
        if self.verbose:
            eng.kv_cache.print_status()
            eng.graph_cache.print_status()
            eng.draft_cache.print_status()
        self.calibrate_aligned()
        self.start(logits)

    def calibrate_aligned(self):
        """Aligned synthetic weights (models/aligned.py) need one calibration of the lm_head's attention read-out once
        a prompt's full and retrieval caches exist; real or random weights: no-op."""
        info = getattr(getattr(getattr(self.eng, "model", None), "weights", None), "aligned", None)
        if info is None or info.get("role") != "target" or "calibration" in info:
            return None
        from ..models import aligned
        return aligned.calibrate_engine(self.ge, self.gamma, self.temperature, self.top_p)

    @torch.inference_mode()
    def start(self, logits):
        probs = norm_logits(logits[:, -1, :], temperature=self.temperature, top_k=self.top_k, top_p=self.top_p)
        self.next_token = int(sample(probs, rng=self.rng))
        if self.verbose:
            spec_stream(self.next_token, self.tokenizer, "cyan")
        self.emitted = [self.next_token]

    @torch.inference_mode()
    def step(self):
        """One outer iteration: Middle_Spec drafting, target verify over the full KV, device-side
        accept/rollback, cache fix-ups.  Returns the number of tokens emitted."""
        eng, ge, gamma, device, bufs, rng = self.eng, self.ge, self.gamma, self.device, self.bufs, self.rng
        tokenizer, verbose = self.tokenizer, self.verbose
        next_token = self.next_token
        n0 = self.n
        # health (TP): a plain load of the pinned mirror of the exchange's error word after EVERY inner record read — after a
        # timed-out exchange the probabilities are NaN-filled, and without the check up to gamma more iterations would draft,
        # append draft KV and write token ids from them before the outer step noticed (advisor, round 4)
        ids, spec_rows, acc_mid = Middle_Spec(next_token, ge, gamma, False, tokenizer, rng=rng, buffers=bufs,
                                              sync_record=self.sync_record, health=self.health)
        self.acc_rate_middle_list.append(acc_mid)
        generated = ids[1:]
        g2 = len(generated)
        self.draft_count += g2
        self.inner_iters += int(round(g2 / (1.0 + acc_mid)))        # g2 = iterations + accepted drafts

        # target model verifies [next, t1..t_g2] against the full KV cache
        rebuild = self.rebuild_every > 0 and (len(self.counts) + 1) % self.rebuild_every == 0
        self.rebuilds += int(rebuild)
        eager = self.eager_every > 0 and (len(self.counts) + 1) % self.eager_every == 0
        fast = None
        # STEP_ON_DEVICE (round 5): the verify reads its tokens from the shared token buffer (the inner graphs left all of
        # [next, t_1 .. t_g2] there) and its positions / slot / key count from device scalars the PREVIOUS step's accept kernel
        # wrote — the chain behind the last inner record is one graph replay, no set-up launch (tf_accept_chain_step)
        on_device = None
        if STEP_ON_DEVICE and self.inner is not None and self.sync_record is None and not self.inclusive_accept and self.top_k <= 0 \
                and not rebuild and not eager and (self.temperature, self.top_p) == (ge.sampling["temperature"], ge.sampling["top_p"]):
            on_device = ge.verify_sets(gamma)
        if on_device is not None:
            if not ge.verify_lengths_current(gamma):
                ge.sync_verify_lengths(gamma)                          # stale (first step, after an eager / rebuild step): one launch each
            tg = ge.target_graphs[len(ids)]
            probs, verify_tokens = tg.replay_in_place(), tg.ids
        elif (HOST_FAST_MASK & 4) and self.top_k <= 0 and not rebuild and not eager and hasattr(ge, "verify_probs_ids") \
                and len(ids) <= 32:
            # captured forward + temperature / top-p: ids, positions and lengths set by ONE launch (ids as kernel arguments)
            fast = ge.verify_probs_ids(ids, self.temperature, self.top_p)
        if on_device is not None:
            pass
        elif fast is not None:
            probs, verify_tokens = fast
        else:
            verify_tokens = bufs.to_device(ids)
            if self.top_k <= 0 and hasattr(ge, "verify_probs"):      # one hipGraph: forward + temperature / top-p
                probs = ge.verify_probs(verify_tokens, self.temperature, self.top_p, rebuild_retrieval=rebuild, eager=eager)
            else:
                logits = ge.inference(input_ids=verify_tokens, rebuild_retrieval=True) if rebuild \
                    else (ge.inference(input_ids=verify_tokens, eager=True) if eager else ge.inference(input_ids=verify_tokens))
                probs = norm_logits(logits[0], temperature=self.temperature, top_k=self.top_k, top_p=self.top_p)
        if getattr(bufs, "rows_generation", None) is not None:
            # spec_rows is a view of the retrieval-verify graph's static output: nothing may have replayed that graph
            # between Middle_Spec's return and this read (the target verify above is a different graph)
            assert ge.verify_generation() == bufs.rows_generation, "retrieval-verify graph replayed before its rows were consumed"
        rec = bufs.chain_out
        rec.arm(4)
        on_cursor = self.inner is not None and self.sync_record is None
        if on_device is not None:
            # ... and leaves the pass tokens in the token buffer and the NEXT verify's scalars (rolled-back length) on the device
            ops.accept_chain_step(probs, spec_rows, ge.tok_buf.view(-1), rng.buf, rng.cursor_tensor(g2 + 1), g2, False, self.eos,
                                  PAD_TOKEN, ge.target_graphs[len(ids)].slot, on_device, rec.tensor)
        elif on_cursor:
            # the uniform stream's device cursor is live (the inner-iteration graphs advance it): the chain reads its numbers
            # behind it and advances it by what it consumed, so no step ever has to re-synchronise the device copy
            ops.accept_chain_cur(probs, spec_rows, verify_tokens.view(-1)[1:], rng.buf, rng.cursor_tensor(g2 + 1), g2,
                                 self.inclusive_accept, self.eos, rec.tensor)
        else:
            ops.accept_chain(probs, spec_rows, verify_tokens.view(-1)[1:], rng.take(g2 + 1), g2, self.inclusive_accept,
                             self.eos, rec.tensor)
        if self.sync_record is not None:
            self.sync_record(rec.tensor)
        count, pred, reason, consumed = rec.read(4)                      # the one host read of the outer step
        if self.health is not None:
            self.health()
        dp = getattr(getattr(eng, "draft", None), "_persist", None)                        # one-launch draft forward: pinned mirror of its error word
        if dp is not None:
            dp.check()
        dev_consumed = consumed
        if self.inclusive_accept and reason == 1 and generated[g2 - 1] == self.eos:
            # TP loop only: an eos accepted as the LAST drafted token ends the loop before the bonus sample
            # (decoding.py:357-360,382-383); the on-chip loop — and tf_accept_chain — go on to the bonus token (:127)
            reason, pred, consumed = 2, self.eos, consumed - 1
        if on_cursor:
            rng.advanced_on_device(dev_consumed)
            if consumed != dev_consumed:                                 # the kernel drew a bonus number the loop does not consume:
                rng.pos -= dev_consumed - consumed                       # step the host position back; the device cursor is
                rng.device_cursor = False                                # re-written before its next reader
        else:
            rng.advance(consumed)
        self.last_reason = reason          # 0 rejection + resample, 1 everything accepted (bonus token), 2 accepted eos

        pass_tokens = [next_token] + generated[:count] + [PAD_TOKEN] * (g2 + 1 - count)
        if reason != 2:
            pass_tokens[count + 1] = pred             # the resampled (:111-118) or the bonus (:127-134) token

        # Launch order (not the reference's statement order; the three pieces touch disjoint buffers): the 68M catch-up
        # forward (:137-139) goes FIRST — ~120 us of device work behind which the host issues the tail copies, the window
        # shift and the next iteration's first launches; issued last, each of those short launches was a host-bound gap
        # (profiles/r04_gap_analysis_decode_steps.txt).
        tok_buf = getattr(ge, "tok_buf", None)
        if on_device is not None:
            ge.dev_len = eng.kv_cache.seq_len - (g2 - count)            # what the accept kernel wrote: the rolled-back length
            ge.replay_draft(g2 + 1)                                      # the pass tokens are in the token buffer already
        elif (HOST_FAST_MASK & 8) and tok_buf is not None and tok_buf.shape[0] == 1 and tok_buf.shape[1] >= len(pass_tokens) \
                and tok_buf.is_cuda and len(pass_tokens) <= 32:
            # straight into the draft graphs' static input (the next Middle_Spec re-initialises it): one launch, no copies
            replay = getattr(ge, "replay_draft", None)
            if ops.HOST_PLANS and replay is not None and getattr(ge, "static_outputs", False) and tok_buf.numel() <= 32:
                if bufs.pass_plan is None:
                    bufs.pass_plan = ops.SetTokensPlan(tok_buf.view(-1))
                bufs.pass_plan(pass_tokens, PAD_TOKEN, n_dst=len(pass_tokens))
                replay(g2 + 1)                                            # (the graph reads tok_buf itself: no input checks)
            else:
                row = tok_buf[:, :len(pass_tokens)]
                ops.set_tokens(row, pass_tokens, PAD_TOKEN)
                ge.graph_draft_inference(input_ids=row, gamma_offset=g2 + 1)
        else:
            ge.graph_draft_inference(input_ids=bufs.to_device(pass_tokens), gamma_offset=g2 + 1)

        eng.kv_cache.seq_len -= (g2 - count)                              # rollback (:124)
        if on_device is None and STEP_ON_DEVICE and self.inner is not None and self.sync_record is None and self.top_k <= 0 \
                and hasattr(ge, "verify_sets") and ge.verify_sets(gamma) is not None:
            # this step set its verify up from the host (eager / rebuild step, first step): bring the device scalars to the
            # rolled-back length HERE, behind the catch-up draft forward, not at the head of the next step's verify chain
            ge.sync_verify_lengths(gamma)
        ge.update_graph_cache()                                           # refresh the retrieval tail (:125)

        self.accepted_count += count
        self.n += count
        self.emitted += generated[:count]
        if verbose:
            for t in generated[:count]:
                spec_stream(t, tokenizer, "green")
        if reason == 0:                                                   # rejection -> residual resample (:111-118)
            self.resample_count += 1
            self.n += 1
            self.emitted.append(pred)
            if verbose:
                spec_stream(pred, tokenizer, "red")
        elif reason == 2:                                                 # accepted eos (:108-110)
            self.draft_count -= g2 - count
        else:                                                             # everything accepted -> bonus token (:127-134)
            self.target_sample_count += 1
            self.n += 1
            self.emitted.append(pred)
            if verbose:
                spec_stream(pred, tokenizer, "blue")
            count += 1
        self.counts.append(count)

        dc = eng.draft_cache
        dc.evict_for_spec(dc.start_size + dc.recent_size + count)
        self.next_token = pred
        return self.n - n0

    def stats(self, seconds):
        acceptance_rate = self.accepted_count / max(self.draft_count, 1)
        return dict(acceptance_rate=acceptance_rate, tokens_per_s=self.n / seconds, tokens=self.emitted, n=self.n,
                    counts=self.counts, accepted=self.accepted_count, drafted=self.draft_count,
                    avg_tokens=acceptance_rate * self.gamma, resampled=self.resample_count,
                    bonus=self.target_sample_count, seconds=seconds, outer_steps=len(self.counts),
                    acc_rate_middle=float(np.mean(self.acc_rate_middle_list)) if self.acc_rate_middle_list else 0.0)


@torch.inference_mode()
def TriForce(tokenizer, graph_engine, input_ids, gamma=4, max_len=256, top_k=-1, top_p=0.9, temperature=0.6, verbose=False,
             file_path=None, dataset=None, spec_args=None, rng=None, return_details=False, rebuild_every=0):
    run = TriForceRunner(tokenizer, graph_engine, gamma, top_k, top_p, temperature, verbose, rng,
                         rebuild_every=rebuild_every)
    run.prefill(input_ids)
    eng, device = run.eng, run.device
    _sync(device)
    time1 = time.time()
    while run.n < max_len:
        run.step()
    _sync(device)
    time2 = time.time()
    st = run.stats(time2 - time1)
    n, acceptance_rate, avg_tokens = st["n"], st["acceptance_rate"], st["avg_tokens"]
    if verbose:
        print(f"Use {time2 - time1} sec to generate {n} tokens (now {eng.kv_cache.seq_len} tokens), "
              f"Tokens/s: {n / (time2 - time1)}", flush=True)
        print(f"accepted rate {acceptance_rate}, avg generated tokens {avg_tokens}")
    if file_path is not None:
        header = "target,acceptance_rate,token/s,avg_tokens,prefill,gen_len,dataset,acc_rate_middle,latency\n"
        entry = (f"{eng.model.config._name_or_path},{acceptance_rate},{n / (time2 - time1)},{avg_tokens},"
                 f"{input_ids.shape[1]},{n},{dataset},{st['acc_rate_middle']},{(time2 - time1) / n}\n")
        if spec_args is not None:
            for k, v in spec_args.items():
                header = header.replace("\n", f",{k}\n")
                entry = entry.replace("\n", f",{v}\n")
        log_csv(file_path, header, entry)
    if return_details:
        return st
    return acceptance_rate, n / (time2 - time1)


################### Dist Spec (TP + offloading) ####################
import types

import torch.distributed as dist


def _bcast_record(t):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=0)


# Who decides.  The reference broadcasts rank 0's samples and its accept draws (decoding.py:230-239, 345-346) because every
# rank draws from its own generator.  Here every rank holds the SAME uniform stream (same seed, explicit numbers) and, behind
# each exchange, bit-identical activations (every rank adds the same partials in the same order: csrc/allreduce.hip, the
# replicated lm_head and draft model do the rest), so every rank reaches rank 0's decision on its own — the path has no
# exchange step for decisions.  TRIFORCE_TP_REPLICATED_DECISIONS=1 runs TriForce_Dist / Middle_Spec_Dist that way (the autoregressive
# baseline and the tree loop keep their one broadcast per token / per level): no broadcast (two per inner
# iteration + one per outer step otherwise), and with the broadcasts the blocking device-to-host reads go — the records travel
# through the pinned mailbox like the single-GPU loop's.  A digest of the emitted stream is compared across the ranks every
# TRIFORCE_TP_REPLICA_CHECK_EVERY outer steps and when a loop ends (ReplicaCheck): a rank that left the common stream raises
# on every rank.  Default off: the broadcast form is the reference's, and neither form has met a second device yet.
# Round 6: the DEFAULT is "auto" — replicate when this group's forwards have been SHOWN bit-identical on every rank at start-up
# (DistributedLlama.replica_litmus: the retrieval forward with all its exchanges on a fixed probe, digests compared across
# the ranks), else broadcast.  "1" forces replicated, "0" forces the broadcast form (the reference's).  Evidence behind the
# default: tests/test_gpu_tp_offload.py — a two-process soak of 2 000+ tokens at T = 1.0 / top-p 0.95 in both exchange forms emits
# the broadcast form's stream on both ranks, and a rank whose uniform stream is knocked out of step fails LOUDLY (stream digest
# or exchange time-out) within the check interval instead of emitting another stream.
TP_REPLICATED_DECISIONS = __import__("os").environ.get("TRIFORCE_TP_REPLICATED_DECISIONS", "auto") == "1"      # (import-time view; see below)
TP_REPLICA_CHECK_EVERY = max(1, int(__import__("os").environ.get("TRIFORCE_TP_REPLICA_CHECK_EVERY", "32")))


def tp_decision_mode():
    v = __import__("os").environ.get("TRIFORCE_TP_REPLICATED_DECISIONS", "auto")
    return {"1": "replicated", "0": "broadcast"}.get(v, "auto")


def tp_sync_record(llm=None):
    """The `sync_record` of the tensor-parallel loops: rank 0's record broadcast, or None when the ranks replicate decisions.
    COLLECTIVE in "auto" mode the first time it sees an engine (the litmus runs one probe forward on every rank)."""
    mode = tp_decision_mode()
    if mode == "replicated":
        return None
    if mode == "broadcast" or llm is None:
        return _bcast_record
    base = getattr(llm, "llm", llm)                           # (a _DistEngine wraps the engine)
    ok = getattr(base, "replica_ok", None)
    if ok is None and hasattr(base, "replica_litmus"):
        ok = base.replica_litmus()
    return None if ok else _bcast_record


class ReplicaCheck:
    """Replicated decisions only: every rank folds the tokens it emitted into a 61-bit digest; every `every` outer steps
    (and on `force`) ONE 4-word MAX all-reduce of (digest, -digest, n, -n) tells every rank whether all ranks hold the same
    stream.  The call sites are functions of the outer step count alone, so ranks that agree meet in the collective; ranks whose
    forwards fell out of step never get here — their exchanges time out first (RuntimeError from `health`)."""

    MOD = (1 << 61) - 1

    def __init__(self, device, every=None):
        self.device, self.every = device, int(every or TP_REPLICA_CHECK_EVERY)
        self.seen = self.digest = self.checks = self.last_step = 0

    def fold(self, tokens):
        d = self.digest
        for t in tokens[self.seen:]:
            d = (d * 1_000_003 + int(t) + 1) % self.MOD
        self.digest, self.seen = d, len(tokens)

    def __call__(self, run, force=False):
        steps = len(run.counts)
        if not (force or steps - self.last_step >= self.every):
            return
        self.last_step = steps
        self.fold(run.emitted)
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        t = torch.tensor([self.digest, -self.digest, run.n, -run.n], dtype=torch.int64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        hi, lo, n_hi, n_lo = t.tolist()
        self.checks += 1
        if hi != -lo or n_hi != -n_lo:
            raise RuntimeError(f"tensor-parallel ranks left the common token stream (rank {dist.get_rank()}: {run.n} tokens, digest "
                               f"{self.digest:#x}; across ranks n in [{-n_lo}, {n_hi}]): replicated decisions need identical "
                               "uniform streams and bit-identical exchanges — unset TRIFORCE_TP_REPLICATED_DECISIONS to broadcast rank 0's")


def sample_dist(probs, rng=None):
    """Rank 0 samples, everyone gets the token (reference decoding.py:230-239) — one 8-byte broadcast, no barrier."""
    tok = sample(probs, rng=rng)
    _bcast_record(tok)
    return tok


class _DistEngine:
    """Presents a DistributedLlama through the GraphInferenceEngine surface the decode loops drive."""

    def __init__(self, llm):
        self.llm = llm
        self._inner = {}
        self.sampling = dict(probs=True, temperature=llm.temperature, top_p=llm.top_p)

        def draft_run(input_ids, gamma_offset=0, probs=True, temperature=0.6, top_p=0.9):
            return llm._draft_run_eager(input_ids, gamma_offset, True, 0.6, 0.9)      # 0.6 / 0.9 hard-wired

        def model_verify(input_ids, position_ids, probs=True, temperature=None, top_p=None):
            # the retrieval-verify forward, issued eagerly (inside a capture: utils/graph_infer._InnerGraphs) — it reads the
            # shared token / position buffers itself; exchanges included, in the order of the whole-forward graph
            return llm._verify_cap["run_all"]()

        self.engine = types.SimpleNamespace(
            model=types.SimpleNamespace(device=llm.device, config=llm.config.model_config),
            kv_cache=llm.kv_cache, graph_cache=llm.retrieval_cache, draft_cache=llm.draft_cache, draft=getattr(llm, "draft", None),
            draft_run=draft_run, model_verify=model_verify)

    # -- round 6: the single-GPU loop's launch structure over the tensor-parallel engine's captured forwards ------------------
    # (what the loops look for on a graph engine: shared static inputs, replay-only draft steps, the one-graph inner iteration,
    #  a target verify that takes its ids as kernel arguments.  All of it needs whole-forward graphs — with segment graphs the
    #  exchanges run eagerly between the stages — and replicated decisions: a broadcast in the middle of an iteration cannot sit
    #  inside its graph.)
    def _whole(self):
        cap = getattr(self.llm, "_verify_cap", None)
        return cap is not None and cap.get("form") == "whole" and "run_all" in cap and bool(getattr(self.llm, "_draft_graphs", None))

    @property
    def tok_buf(self):
        return getattr(self.llm, "tok_buf", None) if self._whole() else None

    @property
    def pos_buf(self):
        return getattr(self.llm, "pos_buf", None) if self._whole() else None

    def replay_draft(self, gamma_offset):
        return self.llm.replay_draft(gamma_offset)

    def verify_probs_ids(self, ids, temperature, top_p):
        return self.llm.verify_probs_ids(ids, temperature, top_p)

    @torch.inference_mode()
    def inner_graphs(self, gamma, rng, record, capture=True):
        """The per-position inner-iteration graphs (graph_infer._InnerGraphs: draft step -> draw -> retrieval verify with its
        exchanges -> accept test, ONE hipGraph per position) over this engine; None when it cannot provide them."""
        import os
        from .graph_infer import _InnerGraphs
        tok = self.tok_buf
        if os.environ.get("TRIFORCE_INNER_GRAPH", "1") == "0" or tok is None or not tok.is_cuda or tok.shape[1] < gamma + 3 \
                or record.numel() < 4 or record.is_cuda or torch.cuda.is_current_stream_capturing():
            return None
        key = _InnerGraphs.key_of(self, gamma, rng, record)
        got = self._inner.get(key)
        if got is None and capture:
            if len(self._inner) >= 2:
                self._inner.pop(next(iter(self._inner)))
            got = self._inner[key] = _InnerGraphs(self, gamma, rng, record)
        return got

    @property
    def static_outputs(self):
        """The captured draft / retrieval-verify forwards can hand out their static output buffers (valid until the same graph
        replays again — what the decode loops need: no 0.9 MB clone and no per-position row copies per inner iteration)."""
        return getattr(self.llm, "_verify_cap", None) is not None and bool(getattr(self.llm, "_draft_graphs", None)) \
            and __import__("os").environ.get("TRIFORCE_TP_STATIC_OUTPUTS", "1") != "0"

    def graph_draft_inference(self, input_ids, gamma_offset=0, clone=True):
        return self.llm.draft_run(input_ids=input_ids, gamma_offset=gamma_offset, clone=clone)   # 0.6/0.9 hard-wired

    def graph_verify(self, input_ids, position_ids, clone=True):
        return self.llm.retrieval_verify(input_ids=input_ids, position_ids=position_ids,
                                         temperature=self.llm.temperature, top_p=self.llm.top_p, clone=clone)

    def verify_generation(self):
        if getattr(self.llm, "_verify_cap", None) is None:
            return None
        return getattr(self.llm, "_verify_gen", 0) + sum(g.generation for g in self._inner.values())

    def inference(self, input_ids, eager=False):
        return self.llm.inference(input_ids=input_ids, eager=eager)

    def update_graph_cache(self):
        self.llm.retrieval_cache.update_graph_cache(self.llm.kv_cache)

    def health(self):
        check = getattr(self.llm, "check_exchange", None)
        if check is not None:
            check("decode step")


@torch.inference_mode()
def Baseline_Dist(tokenizer, graph_engine, input_ids, max_len=256, top_k=-1, top_p=0.9, temperature=0.6, verbose=False,
                  local_rank=0, rng=None):
    """Autoregressive TP baseline (reference decoding.py:243-287): returns (ms per token, generated ids)."""
    llm = graph_engine
    rng = rng or UniformSource(llm.device, seed=1)
    llm.reset()
    logits = llm.prefill(input_ids=input_ids)
    next_token = sample_dist(norm_logits(logits[:, -1, :], temperature=temperature, top_k=top_k, top_p=top_p), rng)
    gen_tokens = torch.zeros((input_ids.size(0), max_len), dtype=torch.long, device=input_ids.device)
    n = 0
    check = getattr(llm, "check_exchange", None)
    _sync(llm.device)
    time1 = time.time()
    while n < max_len:
        logits = llm.inference(input_ids=next_token)
        next_token = sample_dist(norm_logits(logits[:, -1, :], temperature=temperature, top_k=top_k, top_p=top_p), rng)
        gen_tokens[:, n] = next_token.squeeze()
        n += 1
        if check is not None and n % 32 == 0:          # a timed-out exchange NaN-fills its outputs: stop, don't emit
            check("autoregressive step")
    _sync(llm.device)
    time2 = time.time()
    if check is not None:
        check("autoregressive loop")
    return 1000 * (time2 - time1) / n, gen_tokens


@torch.inference_mode()
def Middle_Spec_Dist(next_token, llm, gamma, verbose, tokenizer, rng=None):
    """Reference decoding.py:432-495."""
    ge = llm if isinstance(llm, _DistEngine) else _DistEngine(llm)
    sync = tp_sync_record(ge)
    if sync is None and rng is None:
        # replicated decisions: every rank must hold the SAME uniform stream — an unseeded one differs per rank and the ranks
        # would drift apart silently (in the broadcast form rank 0's draw is everyone's, so None did no harm there).  One
        # seeded stream per engine, continued across calls (advisor, round 5).
        base = ge.llm
        rng = getattr(base, "_replicated_rng", None)
        if rng is None:
            rng = base._replicated_rng = UniformSource(base.device, seed=1)
    return Middle_Spec(next_token, ge, gamma, verbose, tokenizer, rng=rng, sync_record=sync)


@torch.inference_mode()
def TriForce_Dist(tokenizer, llm, input_ids, gamma=4, max_len=256, top_k=-1, top_p=0.9, temperature=0.6, verbose=False,
                  file_path=None, dataset=None, spec_args=None, rng=None, return_details=False):
    """Reference decoding.py:291-428: returns (avg accepted tokens, seconds per token)."""
    ge = _DistEngine(llm)
    rng = rng or UniformSource(llm.device, seed=1)          # same seed on every rank: identical uniform streams
    run = TriForceRunner(tokenizer, ge, gamma, top_k, top_p, temperature, verbose, rng, inclusive_accept=True,
                         sync_record=tp_sync_record(llm))
    run.health = ge.health
    replicas = ReplicaCheck(llm.device) if run.sync_record is None else None
    llm.reset()
    if input_ids.shape[1] != llm.prefill_len:
        # the retrieval cache's chunk grid and the device mirror of the generated rows are laid out for exactly this
        # prompt length (the entry scripts clip / generate prompts to --prefill)
        raise ValueError(f"prompt length {input_ids.shape[1]} != the engine's prefill length {llm.prefill_len}")
    llm.prefill(input_ids=input_ids[:, :-1])
    logits = llm.build_retrieval_cache(input_ids=input_ids[:, -1:])
    info = getattr(llm.weights, "aligned", None)
    if info is not None and "calibration" not in info:        # aligned synthetic weights: one-off read-out calibration
        from ..models import aligned
        aligned.calibrate_llm(llm, gamma, temperature, top_p)
    run.start(logits)
    llm.draft_run(input_ids=input_ids)
    eos = run.eos
    _sync(llm.device)
    time1 = time.time()
    knock = __import__("os").environ.get("TF_TEST_TP_KNOCK_RANK_AT")       # tests only: "rank,step" — that rank skips one uniform there
    while run.n < max_len:
        if knock and [llm.local_rank, len(run.counts)] == [int(x) for x in knock.split(",")]:
            rng.advance(1)
        run.step()
        if replicas is not None:
            replicas(run)
        # the TP loop stops when the token that closed the accept scan — accepted or resampled — is eos
        # (decoding.py:382-383); a bonus token is sampled after that check and never ends the loop
        if run.next_token == eos and run.last_reason != 1:
            break
    if replicas is not None:
        replicas(run, force=True)
    _sync(llm.device)
    time2 = time.time()
    st = run.stats(time2 - time1)
    st["decisions"] = "broadcast" if replicas is None else "replicated"
    st["replica_checks"] = 0 if replicas is None else replicas.checks
    st["inner_graphs"] = run.inner is not None           # one hipGraph per inner iteration (whole-forward graphs + replicated decisions)
    if return_details:
        return st
    return st["avg_tokens"], (time2 - time1) / max(run.n, 1)
