"""Tensor-parallel + KV-offloading engine — host-side mirror of the reference's models/TP_llama.py
(DistributedLlama :27-389, distributed_init :19-25) with the same constructor keywords and methods
(reset / prefill / build_retrieval_cache / draft_run / inference / retrieval_verify / init_parameters).

Parallelism: one process per GPU; attention heads and MLP columns are sharded
Megatron-style (TP_layers.py:126-147), the KV caches, retrieval cache and offloaded KV follow the head
shard; embed, norms, lm_head and the 68M draft are replicated.  The only exchange step is an fp16
all-reduce(SUM) after wo and after down_proj (tensor_op.py:179,326,359) — RCCL over xGMI through
torch.distributed ("nccl" is RCCL on ROCm).

Offloading tier: layers >= on_chip_layers keep their KV in pinned host memory; a target forward streams
them through two device buffers with hipMemcpy2DAsync on a copy stream (tf_kv_h2d_async), the new
tokens' rows go back with tf_kv_d2h_async, and compute/copy are ordered with events — not with the
reference's two device-wide synchronize() per layer (TP_llama.py:222,228).
"""
import math
import os
import socket

import torch
import torch.distributed as dist
import torch.nn.functional as F

from .. import ops
from ..utils.sampling import norm_logits
from .cache import (DistributedKVCacheBuffer, DistributedRetrievalCache, DistributedRetrievalCache_Seqouia,
                    DistributedSimpleCache)
from .config_yarn import LlamaConfig
from .llama_core import LlamaWeights, parse_random_spec, rope_tables_for, softmax_scale_for
from .TP_layers import DistributedOffloadingConfig


def distributed_init(backend=None):
    """torchrun entry: RCCL process group, one device per local rank (reference TP_llama.py:19-25)."""
    backend = backend or os.environ.get("TRIFORCE_DIST_BACKEND", "nccl")
    if backend == "nccl":                       # bind the device first: RCCL communicators attach to the current one
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", 0))))
    if not dist.is_initialized():
        dist.init_process_group(backend=backend)
    return dist.get_rank(), dist.get_world_size()


class TreeMask:
    """Packed tree-attention mask: bit j of ``bits[row0 + i]`` = tree node j visible to query row i."""

    def __init__(self, bits, row0=0):
        self.bits, self.row0 = bits, row0


class DistributedLlama:
    def __init__(self, model_name_or_path: str, dtype=torch.float16, kv_offload=False, on_chip_layers=32, local_rank=0,
                 world_size=1, prefill=32768, bsz=1, gen_len=256, retrieval_budget=4096, retrieval_chunk_size=8, gamma=6,
                 temperature=0.6, top_p=0.9, ssl=0, draft=None, draft_cache=None, flash_attn=True, config=None,
                 device=None, tree_size=0) -> None:
        assert dtype == torch.float16
        self.device = torch.device(device) if device is not None else torch.device("cuda", local_rank)
        self.dtype = dtype
        self.local_rank, self.world_size = local_rank, world_size
        self.kv_offload = kv_offload
        self.ssl, self.flash_attn = ssl, flash_attn
        model_config = config if config is not None else LlamaConfig.from_pretrained(model_name_or_path)
        self.model_name_or_path = model_name_or_path
        self.config = DistributedOffloadingConfig(model_config, local_rank, world_size)
        self.on_chip_layers = min(on_chip_layers, model_config.num_hidden_layers)
        self.vocab_size = model_config.vocab_size
        self.prefill_len = prefill
        self.retrieval_budget = retrieval_budget
        self.temperature, self.top_p, self.gamma, self.bsz = temperature, top_p, gamma, bsz
        self.draft, self.draft_cache = draft, draft_cache
        on_gpu = self.device.type == "cuda"
        self.load_stream = torch.cuda.Stream(device=self.device) if on_gpu else None

        if not kv_offload:
            raise NotImplementedError
        assert bsz == 1
        self.tree_size = tree_size
        budget = prefill + gen_len + 32 + tree_size           # TP_llama_tree.py:70 adds the tree rows
        self.kv_cache = DistributedSimpleCache(self.config, max_budget=budget, device=self.device,
                                               on_chip_layers=self.on_chip_layers, ssl=ssl)
        n_off = model_config.num_hidden_layers - self.on_chip_layers
        self.kv_buffer = [DistributedKVCacheBuffer(self.config, max_budget=budget, device=self.device)
                          for _ in range(2 if n_off > 0 else 0)]
        if retrieval_budget > 0 and tree_size > 0:            # Sequoia tree path (TP_llama_tree.py:72)
            self.retrieval_cache = DistributedRetrievalCache_Seqouia(
                self.config, max_budget=retrieval_budget, device=self.device, prefill=prefill,
                chunk_size=retrieval_chunk_size, tree_size=tree_size)
            self.retrieval_cache.ensure_tail(gen_len + 32 + tree_size)
            self.kv_cache.tail_mirror = self.retrieval_cache
        elif retrieval_budget > 0:
            self.retrieval_cache = DistributedRetrievalCache(self.config, max_budget=retrieval_budget, device=self.device,
                                                             prefill=prefill, chunk_size=retrieval_chunk_size, gamma=gamma)
            self.retrieval_cache.ensure_tail(gen_len + 32)
        else:                                         # autoregressive baseline (offloading_TP.py:75)
            self.retrieval_cache = None

        self.hidden_size = model_config.hidden_size
        self.num_heads = model_config.num_attention_heads
        self.head_dim = self.hidden_size // self.num_heads
        self.num_key_value_heads = model_config.num_key_value_heads
        self.num_key_value_groups = self.num_heads // self.num_key_value_heads
        self.max_position_embeddings = model_config.max_position_embeddings
        if prefill + gen_len > self.max_position_embeddings:
            # the RoPE tables have max_position_embeddings rows; a position beyond them is an out-of-bounds read on the device
            # (found by a 2 000-token soak on a 4 096-position test model: memory access fault) — refuse the configuration
            raise ValueError(f"prefill {prefill} + gen_len {gen_len} exceeds the model's max_position_embeddings "
                             f"{self.max_position_embeddings}: positions past the RoPE tables")
        self.rope_theta = model_config.rope_theta
        self.local_num_heads = self.num_heads // world_size
        self.local_num_key_value_heads = self.num_key_value_heads // world_size
        self.scale = softmax_scale_for(self.head_dim)
        self.tree_scale = 1.0 / math.sqrt(self.head_dim)
        self.weights = None
        self.num_layers = model_config.num_hidden_layers
        self._ev_ready = self._ev_done = None
        self._ar = None                               # utils.oneshot_ar.OneShotAllReduce once enabled (world_size > 1)
        self._xchg = None                             # utils.oneshot_ar.GemmExchange: o_proj / down_proj + exchange, one launch

    # ---------------------------------------------------------------------------------------
    def init_parameters(self, hf_model=None):
        """Shard this rank's slice of the weights (TP_layers.py:126-147).  Accepts an HF-keyed state dict, an
        object with .state_dict(), or None / 'random:<seed>' (random init; every rank draws its own shard)."""
        cfg = self.config.model_config
        W = LlamaWeights(cfg, self.device, rank=self.local_rank, world_size=self.world_size)
        seed = parse_random_spec(hf_model if isinstance(hf_model, str) else self.model_name_or_path) \
            if (hf_model is None or isinstance(hf_model, str)) else None
        from .aligned import parse_spec
        spec = parse_spec(hf_model if isinstance(hf_model, str) else self.model_name_or_path) \
            if (hf_model is None or isinstance(hf_model, str)) else None
        if spec is not None:                              # aligned synthetic weights (models/aligned.py)
            W.init_aligned(spec, "target", attn_keys=max(self.retrieval_budget, 1))
        elif hf_model is None or isinstance(hf_model, str):
            W.init_random(seed if seed is not None else 0)
        else:
            sd = hf_model if isinstance(hf_model, dict) else hf_model.state_dict()
            W.load_state_dict(sd)
        self.weights = W
        cos, sin = rope_tables_for(cfg)
        self.cos_cache, self.sin_cache = cos.to(self.device), sin.to(self.device)
        self.embed_tokens, self.lm_head, self.norm_weight = W.embed, W.lm_head, W.norm
        self.norm_variance_epsilon = W.eps
        # TRIFORCE_ALLREDUCE = auto (one-shot after its self-check, else RCCL) | oneshot (fail if unavailable) | rccl;
        # TRIFORCE_ONESHOT_AR=0 is the older spelling of rccl
        mode = os.environ.get("TRIFORCE_ALLREDUCE", "auto")
        if mode not in ("auto", "oneshot", "rccl"):
            raise ValueError(f"TRIFORCE_ALLREDUCE={mode!r}: expected auto, oneshot or rccl")
        if self.world_size > 1 and self.device.type == "cuda":
            self._detect_shared_device()
        if self.world_size > 1 and self.device.type == "cuda" and mode != "rccl" \
                and os.environ.get("TRIFORCE_ONESHOT_AR", "1") != "0":
            self.enable_oneshot_allreduce()

    def _detect_shared_device(self):
        """COLLECTIVE: how many ranks of this group sit on ONE device (bench.py --share-device, the two-process tests).  The
        one-launch draft forward needs its 256 workgroups co-resident: two ranks' launches on one chip each get part of the CUs
        and wait for the rest until the time-out (found by the 2-rank bench test) — shared devices keep the 13-launch chain."""
        try:
            mine = (socket.gethostname(), str(getattr(torch.cuda.get_device_properties(self.device), "uuid", self.device.index)))
        except Exception:
            mine = (socket.gethostname(), str(self.device.index))
        everyone = [None] * self.world_size
        dist.all_gather_object(everyone, mine)
        self.ranks_per_device = max(everyone.count(e) for e in everyone)
        if self.ranks_per_device > 1:
            # same for the multi-workgroup top-p (rows x 16 resident workgroups per launch): 18-row verifies of two ranks do not
            # fit one chip together
            ops.TOPP_MULTI = False
        draft = getattr(self, "draft", None)
        if self.ranks_per_device > 1 and draft is not None and hasattr(draft, "persist_allowed"):
            draft.persist_allowed = False
            draft._native_key = None                      # (rebuild the native handle without the persistent state)

    # ---------------------------------------------------------------------------------------
    ONESHOT_MAX_ROWS = 32                             # decode-sized blocks; prefill chunks (>= 1 MB) stay on RCCL

    @torch.inference_mode()
    def enable_oneshot_allreduce(self, verbose=False):
        """Switch the decode-sized all-reduces (<= ONESHOT_MAX_ROWS rows x hidden fp16: 8..330 KB) from RCCL to the
        one-shot peer-read kernel (utils/oneshot_ar.py).  Collective.  The kernel is checked here against
        dist.all_reduce on random data — every rank must agree that it was exact (world 2) / within one fp16 rounding of
        the ring's result and that no wait timed out — else the engine stays on RCCL."""
        from ..utils.oneshot_ar import OneShotAllReduce
        forced = os.environ.get("TRIFORCE_ALLREDUCE", "auto")          # rccl | oneshot | auto (bench.py --allreduce)
        ar, why = None, []

        def stage(fn, what):
            """Run one LOCAL stage under try, then let the ranks agree on it: a rank that fails never leaves its peers
            inside a collective it has skipped (every collective below is entered by all ranks or by none)."""
            good = True
            try:
                fn()
            except Exception as ex:
                good = False
                why.append(f"{what}: {type(ex).__name__}: {ex}")
            return self._agree(good)

        def alloc():
            nonlocal ar
            # TRIFORCE_AR_ALTERNATE=1: two staging halves used in turn, no DONE handshake (csrc/allreduce.hip); every
            # forward of this engine issues an even number of exchanges (two per layer), which that form relies on
            ar = OneShotAllReduce(self.local_rank, self.world_size, self.device, self.ONESHOT_MAX_ROWS * self.hidden_size,
                                  connect=False, alternate=os.environ.get("TRIFORCE_AR_ALTERNATE", "0") == "1")

        ok = stage(alloc, "allocation")
        # connect() always reaches its all-gather (a failed export contributes None), and raises afterwards
        ok = ok and stage(lambda: ar.connect(), "handle exchange")
        if ok:
            # self-test: the reference sums come from RCCL first, unconditionally, so the collectives match on every
            # rank whatever happens to the one-shot launches afterwards
            g = torch.Generator(device=self.device).manual_seed(1234 + self.local_rank)
            parts = [torch.randn(rows, self.hidden_size, generator=g, device=self.device).to(torch.float16)
                     for rows in (1, 7, self.ONESHOT_MAX_ROWS)]
            wants = [p.clone() for p in parts]
            for w in wants:
                dist.all_reduce(w, dist.ReduceOp.SUM)

            def selftest():
                for part, want in zip(parts, wants):
                    st = ar.staging(part.shape[0], self.hidden_size)
                    st.copy_(part)
                    got = ar.reduce(st, torch.empty_like(part))
                    torch.cuda.synchronize(self.device)
                    err = (got.float() - want.float()).abs()
                    tol = 0.0 if self.world_size == 2 else 2.0 ** -8 * float(want.float().abs().max())
                    if not bool(torch.isfinite(got).all()) or float(err.max()) > tol:
                        raise RuntimeError(f"self-test mismatch at {part.shape[0]} rows: max err {float(err.max())}")
                ar.check("self-test")

            ok = stage(selftest, "self-test")
        if ok:
            self._ar = ar
            # ... and the fused form of the decode layer's two exchanges: o_proj / down_proj with the all-reduce in their
            # epilogue (TRIFORCE_TP_GEMM_XCHG=0 keeps GEMM -> staging -> exchange kernel)
            if os.environ.get("TRIFORCE_TP_GEMM_XCHG", "1") != "0":
                self._enable_gemm_exchange(stage, why)
        elif ar is not None:
            ar.close()
        self.allreduce_note = "; ".join(why)
        if verbose or self.local_rank == 0:
            form = "one-shot peer reads (xGMI)" + (", alternating staging halves" if ok and ar.alternate else "")
            print(f"[TP] decode all-reduce: {form if ok else 'RCCL'}"
                  + (f" ({self.allreduce_note})" if why else ""), flush=True)
        if not ok and forced == "oneshot":
            raise RuntimeError("TRIFORCE_ALLREDUCE=oneshot but the one-shot all-reduce is unavailable: "
                               + (self.allreduce_note or "a peer rank failed its stage"))
        return ok

    def _enable_gemm_exchange(self, stage, why):
        """Collective: allocate / connect / self-test utils.oneshot_ar.GemmExchange (the ranks agree after every stage, as
        for the exchange kernel); on any failure the engine simply keeps the two-launch form."""
        from ..utils.oneshot_ar import GemmExchange
        xc = None

        def alloc():
            nonlocal xc
            xc = GemmExchange(self.local_rank, self.world_size, self.device, self.ONESHOT_MAX_ROWS * self.hidden_size,
                              connect=False)

        # Ranks that SHARE a device (bench.py --share-device, the two-process tests) run their exchange kernels side by side:
        # every workgroup of the fused form spins on its peers' workgroup for the same panel, so all world x hidden / 16
        # workgroups (8 waves each) must be co-resident or the spinners starve the peers they wait for (advisor, round 4).
        # Up to 768 workgroups fit with margin (256 CUs x 4 of 8 waves); beyond that the two-launch form stays (its exchange
        # kernel caps itself at 64 workgroups).
        try:
            mine = (socket.gethostname(), str(getattr(torch.cuda.get_device_properties(self.device), "uuid", self.device.index)))
        except Exception:
            mine = (socket.gethostname(), str(self.device.index))
        everyone = [None] * self.world_size
        dist.all_gather_object(everyone, mine)
        sharing = max(everyone.count(e) for e in everyone)
        self.ranks_per_device = sharing
        if sharing > 1 and sharing * (self.hidden_size // 16) > 768 and os.environ.get("TRIFORCE_XCHG_SHARED_DEVICE", "0") != "1":
            why.append(f"GEMM+exchange: {sharing} ranks share one device x {self.hidden_size // 16} panels do not fit the chip "
                       "together - two-launch form kept")
            return False
        ok = stage(alloc, "GEMM+exchange allocation")
        ok = ok and stage(lambda: xc.connect(), "GEMM+exchange handle exchange")
        if ok:
            g = torch.Generator(device=self.device).manual_seed(4321 + self.local_rank)
            K = 256
            w = ops.PackedLinear((torch.randn(self.hidden_size, K, generator=g, device=self.device) * 0.05).to(torch.float16))
            acts = [torch.randn(rows, K, generator=g, device=self.device).to(torch.float16) for rows in (1, 7, 18)]
            wants = [ops.linear(a, w) for a in acts]                 # this rank's fp16 partial, summed by RCCL first
            for t in wants:
                dist.all_reduce(t, dist.ReduceOp.SUM)

            def selftest():
                for a, want in zip(acts, wants):
                    for packed in (False, True):
                        x = torch.zeros(a.shape[0], self.hidden_size, dtype=torch.float16, device=self.device)
                        xa = ops.Act.from_rows(x) if packed else x
                        xc.linear_reduce(ops.Act.from_rows(a) if packed else a, w, xa, ops.ss_buffer(self.hidden_size, self.device))
                        torch.cuda.synchronize(self.device)
                        got = xa.rows() if packed else xa
                        err = (got.float() - want.float()).abs()
                        tol = 0.0 if self.world_size == 2 else 2.0 ** -8 * float(want.float().abs().max())
                        if not bool(torch.isfinite(got).all()) or float(err.max()) > tol:
                            raise RuntimeError(f"GEMM+exchange self-test mismatch at {a.shape[0]} rows: {float(err.max())}")
                xc.check("self-test")

            ok = stage(selftest, "GEMM+exchange self-test")
        if ok:
            ok = self._xchg_litmus(xc, stage, why)
        if ok:
            self._xchg = xc
        elif xc is not None:
            xc.close()
        return ok

    def _xchg_litmus(self, xc, stage, why):
        """Which form of the fused exchange this GROUP can trust (verdict round 4, item 4): the fence-free form (default) is
        run through GemmExchange.litmus on the real ranks — TRIFORCE_XCHG_LITMUS_ITERS iterations (default 100 000, 0 skips),
        one rank delayed now and then; a single mismatched element or a time-out on ANY rank switches every rank to the
        fenced form (control blocks reset collectively), which is then run through the same litmus; if that fails too the
        fused form is not used at all.  TRIFORCE_XCHG_FENCE=1 starts with the fenced form.  The verdict goes into
        ``allreduce_note`` / ``xchg_form``."""
        from ..utils.graph_infer import _capture_error_mode
        # (ranks sharing one device — functional runs on a one-GPU box, where every mapping resolves to local HBM — take turns on
        #  the chip and cannot expose a cross-device ordering problem anyway: 5 000 iterations; the dedicated litmus test runs more)
        iters = int(os.environ.get("TRIFORCE_XCHG_LITMUS_ITERS", "5000" if getattr(self, "ranks_per_device", 1) > 1 else "100000"))
        forced = os.environ.get("TRIFORCE_XCHG_FENCE", "0") == "1"
        xc.set_fenced(forced)
        self.xchg_form = "fenced (TRIFORCE_XCHG_FENCE=1)" if forced else "fence-free"
        self.xchg_litmus = []
        if iters <= 0:
            self.xchg_form += ", litmus skipped"
            return True
        for attempt in (0, 1):
            res = {}

            def run():
                res.update(xc.litmus(iters=iters, capture_mode=_capture_error_mode()))
                if res["mismatched_elements"] or res["error_word"]:
                    raise RuntimeError(f"litmus: {res['mismatched_elements']} mismatched elements, error word {res['error_word']} "
                                       f"in {res['iterations']} iterations ({'fenced' if res['fenced'] else 'fence-free'} form)")
            good = stage(run, "GEMM+exchange litmus")
            self.xchg_litmus.append(dict(res, ok=good))
            if good:
                self.xchg_form += f", litmus {res.get('iterations', 0)} iterations clean"
                return True
            if forced or attempt == 1:
                break
            # fence-free form failed somewhere: everyone resets and retries with the fences
            dist.barrier()
            xc.reset()
            dist.barrier()
            xc.set_fenced(True)
            forced = True
            self.xchg_form = "fenced (selected by the litmus: the fence-free form mismatched)"
        dist.barrier()
        xc.reset()
        dist.barrier()
        self.xchg_form = "off (litmus failed in both forms)"
        return False

    def check_exchange(self, where=""):
        """Raise if the one-shot all-reduce ever timed out (its outputs are NaN-filled from then on).  Called once per
        decode step by the loops in utils/decoding.py, after the step's host read."""
        if self._ar is not None:
            self._ar.check(where)
        if self._xchg is not None:
            self._xchg.check(where)

    def reset(self):
        self.kv_cache.reset()
        if self.retrieval_cache is not None:
            self.retrieval_cache.reset()
        if self.draft_cache is not None:
            self.draft_cache.reset()

    # ---------------------------------------------------------------------------------------
    def _all_reduce(self, t):
        if self.world_size > 1:
            dist.all_reduce(t, dist.ReduceOp.SUM)
        return t

    def _partial_out(self, rows, default=None):
        """Where a block's partial o_proj / down_proj output should be written: this rank's one-shot staging buffer for
        decode-sized blocks (the all-reduce then needs no copy), else ``default`` (None = a fresh tensor)."""
        if self._ar is not None and rows <= self.ONESHOT_MAX_ROWS and (rows * self.hidden_size) % 8 == 0:
            return self._ar.staging(rows, self.hidden_size)
        return default

    def _reduce(self, partial, dst=None):
        """Sum ``partial`` over the ranks.  Staged partials go through the one-shot kernel into ``dst`` (a fresh tensor
        when None); anything else is reduced in place by RCCL (and copied to ``dst`` if one is given)."""
        if self.world_size == 1:
            if dst is not None and dst.data_ptr() != partial.data_ptr():
                dst.copy_(partial)
                return dst
            return partial
        if self._ar is not None and self._ar.is_staged(partial):
            return self._ar.reduce(partial, torch.empty_like(partial) if dst is None else dst)
        self._all_reduce(partial)
        if dst is not None and dst.data_ptr() != partial.data_ptr():
            dst.copy_(partial)
            return dst
        return partial

    def _attn_half(self, i, x, d, pos, kl, vl, slot, sk, retrieval_build=False, tree=None, out=None, slot_dev=None,
                   sk_dev=None):
        """Attention block of layer i on this rank's shard up to (not including) the all-reduce: returns the partial
        o_proj output.  x: residual stream (updated in place with the pending MLP output d of the previous layer)."""
        W = self.weights
        Hl, D = W.H_local, W.D
        if d is None:
            h = ops.rmsnorm(x, W.ln1[i], W.eps)
        else:
            h = ops.rmsnorm(d, W.ln1[i], W.eps, residual=x, sum_out=x)
        qkv = ops.linear(h, W.wqkv[i])
        q = ops.rope_append(qkv, self.cos_cache, self.sin_cache, pos, kl, vl, slot, Hl, D, slot0_dev=slot_dev)
        if retrieval_build:                               # tensor_op.py:161-162
            self.retrieval_cache.init_graph_cache((kl, vl), q, i)
        if sk_dev is not None:                            # captured form: slot / key count live in device memory
            a = ops.attn_decode(q, kl, vl, sk, self.scale, sk_dev=sk_dev)
        elif tree is None:
            a = ops.attn_prefill(q, kl, vl, sk, self.scale)
        else:                                             # tensor_op.py:171,265: SDPA, scale 1/sqrt(D) in fp32
            bits, row0, tree_start = tree
            a = ops.attn_tree(q, kl, vl, sk, self.tree_scale, bits, tree_start, mask_row0=row0)
        return ops.linear(a, W.wo[i]) if out is None else ops.linear(a, W.wo[i], out=out)

    def _mlp_half(self, i, x, o, out=None):
        """MLP block of layer i up to the all-reduce: x += o (all-reduced attention output), returns the partial
        down_proj output."""
        W = self.weights
        h = ops.rmsnorm(o, W.ln2[i], W.eps, residual=x, sum_out=x)
        act = ops.mlp_act(h, W.wgu[i])
        return ops.linear(act, W.wd[i]) if out is None else ops.linear(act, W.wd[i], out=out)

    def _layer(self, i, x, d, pos, kl, vl, slot, sk, q_len, retrieval_build=False, tree=None):
        """One decoder layer on this rank's shard.  x: residual stream (updated in place), d: pending MLP output
        of the previous layer (None for layer 0).  Returns the (all-reduced) MLP output of this layer."""
        o = self._reduce(self._attn_half(i, x, d, pos, kl, vl, slot, sk, retrieval_build, tree,
                                         out=self._partial_out(q_len)))                                # tensor_op.py:176-179
        return self._reduce(self._mlp_half(i, x, o, out=self._partial_out(q_len)))                     # tensor_op.py:353-359

    # ---- fused decode layer (<= 32 rows): the single-GPU engine's 5-launch layer with an exchange step in it -------
    # q|k|v GEMM with RMSNorm prologue and RoPE + KV-append epilogue, attention, o_proj, [exchange], gate|up GEMM with
    # RMSNorm prologue and SwiGLU epilogue, down_proj, [exchange].  The exchange (tf_allreduce_oneshot_add_ss) adds the
    # summed partials to the residual stream x in place AND leaves the per-panel sums of squares of the new x, so the
    # norm prologue that follows folds 256 floats per row instead of re-reading x in every workgroup (without that
    # hand-off this form measured SLOWER than the un-fused one: profiles/r02_tp_shard_fused_path_rejected.jsonl).
    # 8 launches per layer instead of 11; used for blocks of <= 16 rows.  At world size 1 the residual / sum-of-squares epilogue of the GEMM itself
    # plays the exchange's part.  TRIFORCE_TP_FUSE=0 keeps the un-fused layer.
    def _fused_decode(self, q_len, tree=None):
        # one 16-row MFMA tile: measured per rank with the exchange kernel in the chain (tools/tp_shard_bench.py
        # --local-exchange, profiles/r02_tp_shard_fused_vs_unfused.jsonl) the fused layer wins at 1 and 7 rows (7B, 8 ranks:
        # retrieval verify 2 001 -> 1 893 us, autoregressive step 4 021 -> 3 333) and loses 1-2 % at the 17 rows of a
        # gamma = 16 verify, where every norm prologue works on two row tiles
        # (with k-octet-major activations — ops.act_packed — the second row tile costs the norm prologue far less; the
        #  row limit of the fused layer is then TRIFORCE_TP_FUSE_MAX_ROWS, default 32)
        limit = int(os.environ.get("TRIFORCE_TP_FUSE_MAX_ROWS", "32" if ops.act_packed(q_len) else "16"))
        if tree is not None or q_len > limit or self.device.type != "cuda":
            return False
        if ops.FUSE_MODE != "all" or os.environ.get("TRIFORCE_TP_FUSE", "1") == "0":
            return False
        if self.world_size > 1 and ((self._ar is None and self._xchg is None) or q_len > self.ONESHOT_MAX_ROWS):
            return False
        W = self.weights
        return (all(isinstance(w, ops.PackedLinear) and w.parts is not None for w in (W.wqkv[0], W.wo[0], W.wgu[0], W.wd[0], W.lm_head))
                and W.wqkv[0].wp_rope is not None and (q_len * self.hidden_size) % 8 == 0)

    def _attn_half_fused(self, i, x, ss, pos, kl, vl, slot, sk, retrieval_build=False, slot_dev=None, sk_dev=None):
        """Attention block of layer i, fused form: returns this rank's partial o_proj output in the exchange's staging
        buffer, or None at world size 1 (x and ss already updated by the GEMM's own epilogue)."""
        W = self.weights
        Hl, D = W.H_local, W.D
        packed = isinstance(x, ops.Act)                   # k-octet-major residual stream (ops.act_packed)
        q = ops.qkv_rope(x, W.wqkv[i], W.ln1[i], W.eps, self.cos_cache, self.sin_cache, pos, kl, vl, slot, Hl, D,
                         slot0_dev=slot_dev, ss_in=ss if i > 0 else None)
        if retrieval_build:                               # tensor_op.py:161-162
            self.retrieval_cache.init_graph_cache((kl, vl), q, i)
        a = ops.attn_decode(q, kl, vl, sk, self.scale, sk_dev=sk_dev, packed=packed)
        if self.world_size == 1:
            ops.linear(a, W.wo[i], resid=x, out=x, ss_out=ss)
            return None
        if self._xchg is not None:                        # o_proj + exchange + residual + sums of squares: one launch
            self._xchg.linear_reduce(a, W.wo[i], x, ss)
            return None
        return ops.linear(a, W.wo[i], out=self._ar.staging(x.shape[0], self.hidden_size, packed=packed))

    def _mlp_half_fused(self, i, x, ss):
        W = self.weights
        act = ops.mlp_act(x, W.wgu[i], ln=W.ln2[i], eps=W.eps, ss_in=ss)
        if self.world_size == 1:
            ops.linear(act, W.wd[i], resid=x, out=x, ss_out=ss)
            return None
        if self._xchg is not None:
            self._xchg.linear_reduce(act, W.wd[i], x, ss)
            return None
        return ops.linear(act, W.wd[i], out=self._ar.staging(x.shape[0], self.hidden_size, packed=isinstance(x, ops.Act)))

    def _exchange_fused(self, part, x, ss):
        """x += sum over ranks of ``part`` (tensor_op.py:179-181,359-360) and ss <- panel sums of squares of the new x."""
        if part is not None:
            self._ar.reduce(part, x, resid=x, ss_out=ss)

    def _layer_fused(self, i, x, ss, pos, kl, vl, slot, sk, retrieval_build=False):
        self._exchange_fused(self._attn_half_fused(i, x, ss, pos, kl, vl, slot, sk, retrieval_build), x, ss)
        self._exchange_fused(self._mlp_half_fused(i, x, ss), x, ss)

    def _finish_fused(self, x, ss):
        W = self.weights
        if W.capture is not None:
            W.capture.append(x.rows() if isinstance(x, ops.Act) else x.clone())
        return ops.linear(x, W.lm_head, out_f32=True, ln=W.norm, eps=W.eps, ss_in=ss).unsqueeze(0)

    def _finish(self, x, d, last_rows=None):
        W = self.weights
        h = ops.rmsnorm(d, W.norm, W.eps, residual=x, sum_out=x)
        if W.capture is not None:
            W.capture.append(x.clone())
        if last_rows is not None and last_rows < h.shape[0]:      # chunked prefill: only the returned rows get logits
            h = h[-last_rows:]
        return ops.linear(h, W.lm_head, out_f32=True).unsqueeze(0)

    @torch.inference_mode()
    def inference(self, input_ids, position_ids=None, attention_mask=None, retrieval_cache=None, eager=False,
                  last_rows=None):
        """Target forward over the full KV cache (TP_llama.py:200-243).  ``eager=True`` skips the captured forward (every
        rank must pass the same value: the eager and the captured forward issue the same exchanges in the same order) —
        bench_tp.py runs every N-th target verify that way so its attention launches can be bracketed by HIP events."""
        W, kvc = self.weights, self.kv_cache
        q_len = input_ids.shape[1]
        S = kvc.seq_len
        tree = None if attention_mask is None else self._tree_mask(attention_mask, S, q_len)
        if S + q_len > kvc.max_budget:
            raise IndexError(f"KV cache overflow: {S}+{q_len} > {kvc.max_budget}")
        cap = getattr(self, "_target_caps", {}).get(q_len)
        if cap is not None and position_ids is None and attention_mask is None and retrieval_cache is None and not eager:
            return self._inference_captured(cap, input_ids)
        if position_ids is None:
            position_ids = (S + torch.arange(q_len, dtype=torch.long, device=self.device)).unsqueeze(0)
        pos = position_ids.reshape(-1).contiguous()
        fused = self._fused_decode(q_len, tree)
        x = ops.embed_rows(self.embed_tokens, input_ids, fused and ops.act_packed(q_len))
        build = retrieval_cache is not None
        n_on, L = self.on_chip_layers, self.num_layers
        tail = self.retrieval_cache if (self.retrieval_cache is not None and S >= self.prefill_len) else None

        if n_on < L:                                      # prime the double buffer (copy stream)
            cs = self.load_stream
            cs.wait_stream(torch.cuda.current_stream(self.device))
            ready = {}
            for idx in (n_on, n_on + 1):
                if idx < L:
                    self.kv_buffer[idx % 2].copy_kv(kvc, idx, cs)
                    ready[idx] = torch.cuda.Event()
                    ready[idx].record(cs)
        d = None
        ss = ops.ss_buffer(self.hidden_size, self.device) if fused else None
        for idx in range(L):
            if idx < n_on:
                kl, vl = kvc.layer_kv(idx)
            else:
                buf = self.kv_buffer[idx % 2]
                torch.cuda.current_stream(self.device).wait_event(ready[idx])      # H2D of this layer landed
                kl, vl = buf.k, buf.v
            if fused:
                self._layer_fused(idx, x, ss, pos, kl, vl, S, S + q_len, retrieval_build=build)
            else:
                d = self._layer(idx, x, d, pos, kl, vl, S, S + q_len, q_len, retrieval_build=build, tree=tree)
            if tail is not None:                          # keep the generated rows on the device for the retrieval tail
                ops.kv_copy_rows_pair(kl.unsqueeze(0), vl.unsqueeze(0), tail.tail_k[idx:idx + 1], tail.tail_v[idx:idx + 1],
                                      S, S - self.prefill_len, q_len)
            if idx >= n_on:
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream(self.device))
                cs.wait_event(done)                       # buffer free + new rows written
                buf.copy_back(kvc, idx, S, q_len, cs)     # D2H of the q_len new rows
                if idx + 2 < L:
                    buf.copy_kv(kvc, idx + 2, cs)          # prefetch into the buffer just released
                    ready[idx + 2] = torch.cuda.Event()
                    ready[idx + 2].record(cs)
        if n_on < L:
            torch.cuda.current_stream(self.device).wait_stream(cs)              # write-backs visible before reuse
        kvc.seq_len = S + q_len
        return self._finish_fused(x, ss) if fused else self._finish(x, d, last_rows)

    def _tree_mask(self, attention_mask, tree_start, q_len):
        """Tree visibility for the block-attention kernel: (bit rows int32, first row, key index of tree column 0).
        Accepts a ``TreeMask`` (bits already packed on the device) or the reference's dense additive mask
        ``(1, 1, q_len, tree_start + T)`` (0 = visible, fp16 min = hidden; SpecTree_TP.py:65-67,170)."""
        if isinstance(attention_mask, TreeMask):
            return attention_mask.bits, attention_mask.row0, tree_start
        m = attention_mask.reshape(q_len, -1)
        assert m.shape[1] >= tree_start, "dense tree mask must cover [prefix | tree]"
        return ops.pack_tree_mask(m[:, tree_start:] == 0), 0, tree_start

    @torch.inference_mode()
    def prefill(self, input_ids):
        from ..utils.graph_infer import chunked_prefill                       # TP_llama.py:246-250
        return chunked_prefill(lambda ids, last_rows=None: self.inference(input_ids=ids, last_rows=last_rows), input_ids)

    @torch.inference_mode()
    def build_retrieval_cache(self, input_ids):
        assert input_ids.shape[-1] == 1
        return self.inference(input_ids=input_ids, retrieval_cache=self.retrieval_cache)

    # ---------------------------------------------------------------------------------------
    # hipGraph capture.  A decode-sized TP forward is ~9 short kernels + 2 RCCL calls per layer: launched eagerly it is
    # host-bound (the reference runs it that way and notes that NCCL could not be captured on its hardware,
    # README.md:58).  Three forms, selected by TRIFORCE_TP_GRAPHS:
    #   "whole"    one graph per forward INCLUDING the RCCL all-reduces (PyTorch-ROCm captures RCCL >= 2.9.6)
    #   "segments" 2L+1 collective-free graphs per forward, the all-reduces issued eagerly between them
    #   "0"        eager
    #   "auto"     (default) world_size == 1: whole.  More ranks: try whole, replay it once against the eager forward
    #              on a probe input, agree across ranks (all-reduce MIN of the verdict) and fall back to segments when
    #              capture raised or the replay disagreed.
    # The target verify is captured too (all layers HBM-resident): its append slot and key count live in device
    # memory (tf_rope_append slot0_dev, tf_attn_decode sk_dev), the launch is sized by the cache capacity.
    # ---------------------------------------------------------------------------------------
    def _stage_buffers(self, q_len, ids=None, pos=None):
        """Static buffers of one captured forward.  ``ids`` / ``pos``: capture over these views instead of own buffers — the
        retrieval verify reads the engine's shared token / position buffers, which the decode loop's kernels write directly
        (round 6: the tensor-parallel loop runs the single-GPU loop's launch structure)."""
        dev, hid = self.device, self.hidden_size
        return dict(ids=torch.zeros((1, q_len), dtype=torch.long, device=dev) if ids is None else ids,
                    pos=torch.arange(q_len, device=dev, dtype=torch.long) if pos is None else pos,
                    base=torch.arange(q_len, device=dev, dtype=torch.long),
                    slot=torch.zeros(1, dtype=torch.int32, device=dev),
                    sk=torch.full((1,), q_len, dtype=torch.int32, device=dev),
                    x=(ops.Act(torch.zeros(hid // 8, q_len, 8, dtype=torch.float16, device=dev), q_len)
                       if (self._fused_decode(q_len) and ops.act_packed(q_len))
                       else torch.zeros(q_len, hid, dtype=torch.float16, device=dev)),
                    o=torch.zeros(q_len, hid, dtype=torch.float16, device=dev),
                    d=torch.zeros(q_len, hid, dtype=torch.float16, device=dev),
                    ss=torch.zeros(hid // 16, 32, dtype=torch.float32, device=dev))

    def _stages(self, st, kind):
        """The forward as a list of collective-free stages [(fn, buffer to all-reduce afterwards | None)] on static
        buffers.  kind "retrieval": gamma+1 tokens over the retrieval cache -> probabilities;  kind "target": a verify
        block over the full cache at the device-resident length -> logits."""
        L = self.num_layers
        rc, kvc = self.retrieval_cache, self.kv_cache

        q_len = st["x"].shape[0]
        if self._fused_decode(q_len):
            return self._stages_fused(st, kind)
        # Where a stage writes its partial is resolved WHEN THE STAGE RUNS (or is captured), and each stage hands it to
        # its own exchange step through its own cell: with alternating staging halves (TRIFORCE_AR_ALTERNATE=1)
        # consecutive exchanges read different halves, so two partials taken from staging() up front would share one.

        def attn(i, cell):
            def run():
                if i == 0:
                    st["x"].copy_(self.embed_tokens[st["ids"].reshape(-1)])
                d = None if i == 0 else st["d"]
                part = cell["p"] = self._partial_out(q_len, st["o"])          # staging, or in place
                if kind == "retrieval":
                    kl, vl = rc.layer_kv(i)
                    self._attn_half(i, st["x"], d, st["pos"], kl, vl, rc.spec_slot, rc.real_budget, out=part)
                else:
                    kl, vl = kvc.layer_kv(i)
                    self._attn_half(i, st["x"], d, st["pos"], kl, vl, 0, kvc.max_budget, out=part,
                                    slot_dev=st["slot"], sk_dev=st["sk"])
                return part
            return run

        def mlp(i, cell):
            def run():
                part = cell["p"] = self._partial_out(q_len, st["d"])
                return self._mlp_half(i, st["x"], st["o"], out=part)
            return run

        def finish():
            logits = self._finish(st["x"], st["d"])
            if kind == "retrieval":
                return norm_logits(logits[0], temperature=self.temperature, top_k=-1, top_p=self.top_p)
            return logits

        stages = []                                   # (collective-free stage, its exchange step | None)
        for i in range(L):
            ca, cm = {}, {}
            stages.append((attn(i, ca), lambda c=ca: self._reduce(c["p"], st["o"])))
            stages.append((mlp(i, cm), lambda c=cm: self._reduce(c["p"], st["d"])))
        stages.append((finish, None))
        return stages

    def _stages_fused(self, st, kind):
        """``_stages`` for the fused decode layer: the exchange steps update st["x"] / st["ss"] in place."""
        L = self.num_layers
        rc, kvc = self.retrieval_cache, self.kv_cache
        x, ss = st["x"], st["ss"]

        def attn(i, cell):                                # cell: partial handed from this stage to ITS exchange step
            def run():
                if i == 0:
                    ops.embed_rows(self.embed_tokens, st["ids"], isinstance(x, ops.Act), out=x)
                if kind == "retrieval":
                    kl, vl = rc.layer_kv(i)
                    cell["p"] = self._attn_half_fused(i, x, ss, st["pos"], kl, vl, rc.spec_slot, rc.real_budget)
                else:
                    kl, vl = kvc.layer_kv(i)
                    cell["p"] = self._attn_half_fused(i, x, ss, st["pos"], kl, vl, 0, kvc.max_budget,
                                                      slot_dev=st["slot"], sk_dev=st["sk"])
                return x
            return run

        def mlp(i, cell):
            def run():
                cell["p"] = self._mlp_half_fused(i, x, ss)
                return x
            return run

        def finish():
            logits = self._finish_fused(x, ss)
            if kind == "retrieval":
                return norm_logits(logits[0], temperature=self.temperature, top_k=-1, top_p=self.top_p)
            return logits

        multi = self.world_size > 1
        stages = []
        for i in range(L):
            ca, cm = {}, {}
            stages.append((attn(i, ca), (lambda c=ca: self._exchange_fused(c["p"], x, ss)) if multi else None))
            stages.append((mlp(i, cm), (lambda c=cm: self._exchange_fused(c["p"], x, ss)) if multi else None))
        stages.append((finish, None))
        return stages

    def _capture_forward(self, q_len, kind, form):
        from ..utils.graph_infer import _capture
        tok = getattr(self, "tok_buf", None)
        if kind == "retrieval" and tok is not None and tok.shape[1] >= q_len and self.pos_buf.numel() == q_len:
            st = self._stage_buffers(q_len, ids=tok[:, :q_len], pos=self.pos_buf.view(-1))
        else:
            st = self._stage_buffers(q_len)
        stages = self._stages(st, kind)
        if form == "whole":
            def run_all():
                out = None
                for fn, exchange in stages:
                    out = fn()
                    if exchange is not None:
                        exchange()
                return out
            graph, out = _capture(run_all, (), self._mempool, 3)
            return dict(form="whole", graph=graph, out=out, st=st, T=self.temperature, P=self.top_p, run_all=run_all)
        # segments: the exchanges run eagerly BETWEEN the stage graphs at replay time, so none is issued while the stages
        # are captured one after the other.  With alternating staging halves every stage must still be captured against
        # the half its exchange will read at replay: count the (not yet issued) exchanges by hand while capturing and put
        # the count back afterwards; a replay must then start at the same parity (checked in _replay).
        ar = self._ar
        base = ar._issued if ar is not None else 0
        graphs = []
        out = None
        try:
            for fn, exchange in stages:
                g, out = _capture(fn, (), self._mempool, 2)
                graphs.append((g, exchange))
                if ar is not None and exchange is not None:
                    ar._issued += 1
        finally:
            if ar is not None:
                ar._issued = base
        return dict(form="segments", graphs=graphs, out=out, st=st, T=self.temperature, P=self.top_p,
                    parity=(base & 1) if (ar is not None and ar.alternate) else None)

    def _replay(self, cap, clone=True):
        if cap["form"] == "whole":
            cap["graph"].replay()
        else:
            if cap.get("parity") is not None and (self._ar._issued & 1) != cap["parity"]:
                raise RuntimeError("alternating one-shot all-reduce: this forward's stage graphs were captured at the other "
                                   "exchange parity (an odd number of exchanges ran since) — re-run initialize_graphs()")
            for g, exchange in cap["graphs"]:
                g.replay()
                if exchange is not None:
                    exchange()
        return cap["out"].clone() if clone else cap["out"]

    def _agree(self, ok):
        """Same verdict on every rank (a capture that failed on one rank must be dropped by all)."""
        if self.world_size == 1:
            return ok
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device)
        dist.all_reduce(t, dist.ReduceOp.MIN)
        return bool(t.item())

    @torch.inference_mode()
    def _try_whole(self, q_len, kind, verbose):
        """Capture a whole-forward graph and check one replay against the eager forward on a probe input."""
        cap, ok = None, True
        try:
            cap = self._capture_forward(q_len, kind, "whole")
            ids = torch.full((1, q_len), 7, dtype=torch.long, device=self.device)
            if kind == "retrieval":
                S = self.kv_cache.seq_len
                pos = torch.arange(S, S + q_len, device=self.device).unsqueeze(0)
                want = norm_logits(self.retrieval_inference(ids, pos)[0], temperature=self.temperature, top_k=-1,
                                   top_p=self.top_p)
                cap["st"]["ids"].copy_(ids)
                cap["st"]["pos"].copy_(pos.reshape(-1))
                got = self._replay(cap)
                ok = bool(torch.isfinite(got).all()) and float((got - want).abs().max()) < 5e-2
            else:
                got = self._inference_captured(cap, ids, advance=False)
                ok = bool(torch.isfinite(got).all())
            torch.cuda.synchronize(self.device)
        except Exception as ex:                            # capture of a collective refused / failed: use segments
            ok = False
            if self._ar is not None:
                try:
                    self._ar.resync()                       # the failed capture may have counted an odd number of exchanges
                except Exception:
                    pass
            if verbose or self.local_rank == 0:
                print(f"[TP graphs] whole-forward capture ({kind}, q={q_len}) unavailable: {type(ex).__name__}: {ex}",
                      flush=True)
        return cap if self._agree(ok) else None

    @torch.inference_mode()
    def initialize_graphs(self, gamma=None, capture_verify=None, verbose=False):
        from ..utils.graph_infer import _capture
        gamma = self.gamma if gamma is None else gamma
        mode = os.environ.get("TRIFORCE_TP_GRAPHS", "auto")
        mode = {"1": "whole", "": "auto"}.get(mode, mode)
        if capture_verify is not None:
            mode = "whole" if capture_verify else "0"
        if self._ar is not None:          # host exchange count <- device epoch (a discarded capture or a forward that
            self._ar.resync()             # raised midway may have counted exchanges the device never ran)
        self._mempool = torch.cuda.graphs.graph_pool_handle()
        self._draft_graphs = {}
        # ONE token buffer is the static input of every draft graph (its first gamma_offset + 1 entries) and of the
        # retrieval-verify forward (its first gamma + 1, with ``pos_buf``), like the single-GPU engine's (utils/graph_infer.py)
        self.tok_buf = torch.zeros((1, gamma + 3), dtype=torch.long, device=self.device)
        self.pos_buf = torch.arange(gamma + 1, device=self.device).unsqueeze(0).clone()
        for off in range(gamma + 3):                       # replicated 68M draft steps: no collective inside
            ids = self.tok_buf[:, :off + 1]
            graph, out = _capture(lambda t, off=off: self._draft_run_eager(t, off, True, 0.6, 0.9), (ids,),
                                  self._mempool, 3)
            self._draft_graphs[off] = (graph, ids, out)
        self._verify_cap, self._target_caps = None, {}
        self.graph_form = "eager"
        if self.retrieval_cache is None or mode == "0":
            self.reset()
            return
        target_ok = self.on_chip_layers == self.num_layers
        jobs = [(gamma + 1, "retrieval")] + ([(gamma + 1, "target"), (gamma + 2, "target")] if target_ok else [])
        for q_len, kind in jobs:
            cap = None
            if mode == "whole" or (mode == "auto" and self.world_size == 1):
                cap = self._capture_forward(q_len, kind, "whole")
            elif mode == "auto":
                cap = self._try_whole(q_len, kind, verbose)
                if cap is None:                            # do not retry the collective capture for the other shapes,
                    mode = "segments-auto"                 # and start from a fresh graph memory pool
                    self._mempool = torch.cuda.graphs.graph_pool_handle()
            if cap is None and mode == "segments":
                cap = self._capture_forward(q_len, kind, "segments")
            elif cap is None and mode == "segments-auto":  # fallback of the fallback: eager
                ok = True
                try:
                    cap = self._capture_forward(q_len, kind, "segments")
                except Exception as ex:
                    ok = False
                    if self.local_rank == 0:
                        print(f"[TP graphs] segment capture ({kind}, q={q_len}) failed: {type(ex).__name__}: {ex}", flush=True)
                if not self._agree(ok):
                    cap, mode = None, "0"
            if mode == "0":
                break
            if kind == "retrieval":
                self._verify_cap = cap
            elif cap is not None:
                self._target_caps[q_len] = cap
        self.graph_form = self._verify_cap["form"] if self._verify_cap else "eager"
        self.reset()
        # the probe replays above ran the one-shot all-reduce inside the captured forwards: if any rank saw a peer
        # wait time out there, every rank drops to RCCL and captures again — before any real state exists
        healthy = (self._ar is None or self._ar.error() == 0) and (self._xchg is None or self._xchg.error() == 0)
        if (self._ar is not None or self._xchg is not None) and not self._agree(healthy):
            if verbose or self.local_rank == 0:
                print("[TP] one-shot all-reduce timed out inside a captured forward: falling back to RCCL", flush=True)
            for obj in (self._ar, self._xchg):
                if obj is not None:
                    obj.close()
            self._ar = self._xchg = None
            return self.initialize_graphs(gamma, capture_verify, verbose)

    def _inference_captured(self, cap, input_ids, advance=True):
        kvc, S, q_len = self.kv_cache, self.kv_cache.seq_len, input_ids.shape[1]
        st = cap["st"]
        st["ids"].copy_(input_ids)
        torch.add(st["base"], S, out=st["pos"])
        st["slot"].fill_(S)
        st["sk"].fill_(S + q_len)
        out = self._replay(cap)
        if not advance:
            return out
        tail = self.retrieval_cache
        if tail is not None and S >= self.prefill_len:    # device mirror of the generated rows, all layers at once
            ops.kv_copy_rows_pair(kvc.k, kvc.v, tail.tail_k, tail.tail_v, S, S - self.prefill_len, q_len)
        kvc.seq_len = S + q_len
        return out

    @torch.inference_mode()
    def replica_litmus(self, rounds=2):
        """COLLECTIVE start-up check behind replicated decisions (utils/decoding.tp_sync_record, "auto"): every rank runs the
        retrieval forward — every layer's two exchanges, the replicated lm_head — on the same fixed probe tokens and folds
        the fp32 bit patterns of the logits into two weighted sums; one 4-word MAX all-reduce of (s1, -s1, s2, -s2) says
        whether every rank holds the SAME bits.  Only then may each rank take rank 0's decisions on its own.  Cached in
        ``replica_ok``; a single rank is trivially true; any failure of the probe counts as "not shown"."""
        if self.world_size == 1:
            self.replica_ok = True
            return True
        ok = True
        try:
            g = self.gamma
            S = self.kv_cache.seq_len
            ids = ((torch.arange(g + 1, device=self.device) * 37 + 11) % self.vocab_size).view(1, -1)
            pos = torch.arange(S, S + g + 1, device=self.device).unsqueeze(0)
            s1 = s2 = 0
            for r in range(rounds):
                bits = self.retrieval_inference(ids, pos).float().contiguous().view(torch.int32).reshape(-1).to(torch.int64)
                w = (torch.arange(bits.numel(), device=self.device, dtype=torch.int64) % 1021) + 1
                s1 = (s1 * 31 + int(bits.sum())) % ((1 << 61) - 1)
                s2 = (s2 * 31 + int((bits * w).sum() % ((1 << 61) - 1))) % ((1 << 61) - 1)
            t = torch.tensor([s1, -s1, s2, -s2], dtype=torch.int64, device=self.device)
        except Exception as ex:                            # a probe that cannot run shows nothing
            ok = False
            t = torch.zeros(4, dtype=torch.int64, device=self.device)
            if self.local_rank == 0:
                print(f"[TP] replica litmus could not run ({type(ex).__name__}: {ex}): decisions are broadcast", flush=True)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        hi1, lo1, hi2, lo2 = t.tolist()
        same = hi1 == -lo1 and hi2 == -lo2
        self.replica_ok = bool(self._agree(ok and same))
        if self.local_rank == 0:
            print(f"[TP] replica litmus: the ranks' retrieval-forward logits are {'bit-identical' if self.replica_ok else 'NOT identical'}"
                  f" -> decisions {'replicated (no record broadcasts)' if self.replica_ok else 'broadcast from rank 0'}", flush=True)
        return self.replica_ok

    @torch.inference_mode()
    def verify_probs_ids(self, ids, temperature, top_p):
        """Target verify of a python list of token ids through the captured forward of that length: ids, positions and the two
        length scalars set by ONE launch (tf_set_tokens: the ids travel as kernel arguments), replay, temperature / top-p.
        (probabilities, the forward's (1, q_len) device token row), or None when there is no whole-forward graph of that length."""
        cap = getattr(self, "_target_caps", {}).get(len(ids))
        if cap is None or cap["form"] != "whole" or len(ids) > 32:
            return None
        kvc, S, q_len, st = self.kv_cache, self.kv_cache.seq_len, len(ids), cap["st"]
        if cap.get("plan") is None:
            cap["plan"] = ops.SetTokensPlan(st["ids"].view(-1), st["pos"], st["slot"], st["sk"])
        cap["plan"](list(ids), 0, pos0=S, sk_val=S + q_len)
        logits = self._replay(cap, clone=False)
        tail = self.retrieval_cache
        if tail is not None and S >= self.prefill_len:    # device mirror of the generated rows, all layers at once
            ops.kv_copy_rows_pair(kvc.k, kvc.v, tail.tail_k, tail.tail_v, S, S - self.prefill_len, q_len)
        kvc.seq_len = S + q_len
        return norm_logits(logits[0], temperature=temperature, top_k=-1, top_p=top_p), st["ids"]

    @torch.inference_mode()
    def draft_run(self, input_ids, gamma_offset: int = 0, probs=True, temperature=0.6, top_p=0.9, clone=True):
        """Replicated 68M draft (TP_llama.py:117-132).  NB the reference's call sites never pass temperature /
        top_p, so the draft always samples at 0.6 / 0.9 — kept."""
        g = getattr(self, "_draft_graphs", None)
        if g and probs and input_ids.shape[-1] <= 64 and gamma_offset in g and (temperature, top_p) == (0.6, 0.9) \
                and input_ids.shape[-1] == gamma_offset + 1:
            graph, ids, out = g[gamma_offset]
            if input_ids.data_ptr() != ids.data_ptr():
                ids.copy_(input_ids)
            graph.replay()
            return out.clone() if clone else out           # (clone=False: valid until THIS graph replays again)
        return self._draft_run_eager(input_ids, gamma_offset, probs, temperature, top_p)

    def replay_draft(self, gamma_offset):
        """``draft_run(tok_buf[:, :gamma_offset + 1], gamma_offset, clone=False)`` for a caller that has written the tokens into
        ``tok_buf`` itself: the replay and nothing else."""
        graph, _, out = self._draft_graphs[gamma_offset]
        graph.replay()
        return out

    def _draft_run_eager(self, input_ids, gamma_offset, probs, temperature, top_p):
        if input_ids.shape[-1] > 64:
            for i in range(math.ceil(input_ids.shape[1] / 128)):
                self.draft_cache.evict_prefill(128)
                logits = self.draft(input_ids=input_ids[:, i * 128:(i + 1) * 128], kv_cache=self.draft_cache,
                                    graph_cache=None).logits
        else:
            out = self.draft.forward(input_ids, self.draft_cache, self.draft_cache, gamma_offset,
                                     probs=(temperature, top_p) if probs else None)
            return out.probs if probs else out.logits
        if probs:
            return norm_logits(logits[0, -1:], temperature=temperature, top_k=-1, top_p=top_p)[0]
        return logits

    @torch.inference_mode()
    def retrieval_inference(self, input_ids, position_ids):
        """Retrieval-cache (spec) forward: gamma+1 tokens against the B+gamma+1 retrieval slots (TP_llama.py:371-385)."""
        W, rc = self.weights, self.retrieval_cache
        q_len = input_ids.shape[1]
        assert q_len == rc.gamma + 1
        pos = position_ids.reshape(-1).contiguous()
        fused = self._fused_decode(q_len)
        x = ops.embed_rows(self.embed_tokens, input_ids, fused and ops.act_packed(q_len))
        if fused:
            ss = ops.ss_buffer(self.hidden_size, self.device)
            for idx in range(self.num_layers):
                kl, vl = rc.layer_kv(idx)
                self._layer_fused(idx, x, ss, pos, kl, vl, rc.spec_slot, rc.real_budget)
            return self._finish_fused(x, ss)
        d = None
        for idx in range(self.num_layers):
            kl, vl = rc.layer_kv(idx)
            d = self._layer(idx, x, d, pos, kl, vl, rc.spec_slot, rc.real_budget, q_len)
        return self._finish(x, d)

    @torch.inference_mode()
    def retrieval_verify(self, input_ids, position_ids, temperature=0.6, top_p=0.9, clone=True):
        cap = getattr(self, "_verify_cap", None)
        if cap is not None and (temperature, top_p) == (cap["T"], cap["P"]):
            if input_ids.data_ptr() != cap["st"]["ids"].data_ptr():
                cap["st"]["ids"].copy_(input_ids)
            if position_ids.data_ptr() != cap["st"]["pos"].data_ptr():
                cap["st"]["pos"].copy_(position_ids.reshape(-1))
            self._verify_gen = getattr(self, "_verify_gen", 0) + 1      # lifetime token of a static output handed out
            return self._replay(cap, clone=clone)
        logits = self.retrieval_inference(input_ids, position_ids)
        return norm_logits(logits[0], temperature=temperature, top_k=-1, top_p=top_p)
