"""Draft model (JackFram/llama-68m shape) on the HIP ops — mirror of the reference's
models/modeling_llama_68m.py: keys are cached UN-rotated in a StreamingLLM cache and RoPE is
re-applied to every cached key on read with cache-relative positions (:151-178); a decode call with
``gamma_offset=n`` recomputes all n+1 speculative tokens (:151-162).
"""
import torch

from .. import ops
from .config_yarn import LlamaConfig
from .llama_core import (CausalLMOutput, LlamaWeights, load_checkpoint_state_dict, parse_random_spec,
                         rope_tables_plain, softmax_scale_for)


class LlamaForCausalLM:
    def __init__(self, config: LlamaConfig, device="cuda:0"):
        self.config = config
        self.device = torch.device(device)
        self.dtype = torch.float16
        self.weights = LlamaWeights(config, self.device)
        D = config.hidden_size // config.num_attention_heads
        cos, sin = rope_tables_plain(D, config.max_position_embeddings, config.rope_theta)   # 68m.py:123-128
        self.cos, self.sin = cos.to(self.device), sin.to(self.device)
        self.scale = softmax_scale_for(D)
        self.vocab_size = config.vocab_size
        # the one-launch forward (ops.DraftPersist) needs the whole chip to itself while it runs: an engine whose ranks share a
        # device switches it off (models/TP_llama._detect_shared_device)
        self.persist_allowed = True

    @classmethod
    def from_pretrained(cls, name_or_path, torch_dtype=torch.float16, device_map="cuda:0", config=None, **_):
        assert torch_dtype == torch.float16
        seed = parse_random_spec(name_or_path)
        if seed is not None:
            assert config is not None, "random:<seed> needs config="
            return cls(config, device_map).init_random(seed)
        from .aligned import parse_spec
        spec = parse_spec(name_or_path)
        if spec is not None:                                # aligned[:draft_acc[:retrieval_acc[:seed]]]
            assert config is not None, "aligned:... needs config="
            return cls(config, device_map).init_aligned(spec, attn_keys=_.get("attn_keys", 256))
        cfg = config or LlamaConfig.from_pretrained(name_or_path)
        m = cls(cfg, device_map)
        m.weights.load_state_dict(load_checkpoint_state_dict(name_or_path))
        return m

    @classmethod
    def from_state_dict(cls, config, sd, device="cuda:0"):
        m = cls(config, device)
        m.weights.load_state_dict(sd)
        return m

    def init_random(self, seed):
        self.weights.init_random(seed)
        return self

    def init_aligned(self, spec, attn_keys=256):
        """Aligned synthetic weights (models/aligned.py), this model in the draft role."""
        self.weights.init_aligned(spec, "draft", attn_keys=attn_keys)
        return self

    def eval(self):
        return self

    @torch.inference_mode()
    def __call__(self, input_ids, kv_cache=None, graph_cache=None, position_ids=None, gamma_offset=-1,
                 attention_mask=None, storage_ids=None):
        return self.forward(input_ids, kv_cache, graph_cache, gamma_offset)

    # ---- native chain (tf_draft_forward_68m): one C call issues the whole forward --------------------------------
    def _native_model(self):
        """TfDraftModel over this model's packed weights, or None (CPU tensors / a weight the fused kernels cannot
        take / TRIFORCE_FUSE != all / TRIFORCE_DRAFT_NATIVE=0).  Rebuilt when the weight tensors are replaced (aligned
        re-calibration and in-place re-draws keep their storage)."""
        W = self.weights
        key = (W.embed.data_ptr(), id(W.lm_head), id(W.wqkv[0]))        # a reloaded state dict brings new tensors
        if getattr(self, "_native_key", None) != key:
            import os
            ok = (self.device.type == "cuda" and ops.FUSE_MODE == "all" and os.environ.get("TRIFORCE_DRAFT_NATIVE", "1") != "0"
                  and isinstance(W.lm_head, ops.PackedLinear))
            self._native = ops.draft_model_struct(W.embed, W.ln1, W.wqkv, W.wo, W.ln2, W.wgu, W.wd, W.norm, W.lm_head,
                                                  self.cos, self.sin, W.H, W.D, W.eps, self.scale) if ok else None
            # the one-launch form's control block + workspace (ops.DraftPersist) when the shape is the 68M draft's
            self._persist = None
            if self._native is not None and ops.DRAFT_PERSIST and self.persist_allowed and ops.draft_persist_supported(self._native, 1, 1):
                self._persist = ops.DraftPersist(self._native, self.device)
            self._native_key = key
        return self._native

    def _native_cache(self, c):
        key = id(c)
        if getattr(self, "_native_cache_key", None) != key:
            self._native_cache_struct, self._native_cache_key = ops.draft_cache_struct(c, self.weights.L), key
            self._native_cache_ref = c                                  # keeps id(c) from being recycled
        return self._native_cache_struct

    def forward(self, input_ids, kv_cache, graph_cache=None, gamma_offset=-1, probs=None):
        """``probs`` = (temperature, top_p): also return the top-p probability row of the last token (out.probs)."""
        W = self.weights
        H, D = W.H, W.D
        q_len = input_ids.shape[1]
        spec = gamma_offset >= 0
        native = self._native_model() if q_len <= ops.SKINNY_MAX_ROWS else None
        if native is not None:
            if spec:
                c = graph_cache
                assert q_len == gamma_offset + 1 and q_len <= c.gamma + 3
                slot0, kv_len = c.spec_slot, gamma_offset + c.start_size + c.recent_size + 1
            else:
                c = kv_cache
                slot0 = c.seq_len
                kv_len = slot0 + q_len
                for i in range(W.L):
                    c.append_slot(i, q_len)
            logits, p = ops.draft_forward(native, self._native_cache(c), input_ids.reshape(-1), slot0, kv_len, probs,
                                          persist=self._persist)
            out = CausalLMOutput(logits.unsqueeze(0))
            out.probs = p
            return out
        x = W.embed[input_ids.reshape(-1)]
        if spec:                                                          # 68m.py:151-162
            c = graph_cache
            assert q_len == gamma_offset + 1 and q_len <= c.gamma + 3
            slot0 = c.spec_slot
            kv_len = gamma_offset + c.start_size + c.recent_size + 1
            pos = torch.arange(slot0, slot0 + q_len, dtype=torch.long, device=self.device)
        else:                                                             # 68m.py:164-178, :286-291
            c = kv_cache
            slot0 = c.seq_len
            kv_len = slot0 + q_len
            pos = torch.arange(slot0, slot0 + q_len, dtype=torch.long, device=self.device)
        mode = ops.FUSE_MODE if (ops.can_fuse(x, W.wqkv[0], W.wo[0], W.wgu[0], W.wd[0], W.lm_head)
                                 and W.wqkv[0].wp_rope is not None) else "none"
        fused = mode in ("all", "all2")
        ss = ops.ss_buffer(x.shape[1], x.device) if mode == "all" else None     # sum(x^2) hand-off between GEMMs
        d = None
        for i in range(W.L):
            kl, vl = c.layer_kv(i)
            if not spec:
                c.append_slot(i, q_len)
            if fused:
                q = ops.qkv_rope(x, W.wqkv[i], W.ln1[i], W.eps, self.cos, self.sin, pos, kl, vl, slot0, H, D,
                                 rotate_k=False, ss_in=ss if i > 0 else None)
            else:
                if d is None:
                    h = ops.rmsnorm(x, W.ln1[i], W.eps)
                else:
                    h = ops.rmsnorm(d, W.ln1[i], W.eps, residual=x, sum_out=x)
                if mode == "rope":
                    q = ops.qkv_rope(h, W.wqkv[i], None, 0.0, self.cos, self.sin, pos, kl, vl, slot0, H, D,
                                     rotate_k=False)
                else:
                    q = ops.rope_append(ops.linear(h, W.wqkv[i]), self.cos, self.sin, pos, kl, vl, slot0, H, D,
                                        rotate_k=False)
            a = ops.attn_rope_on_read(q, kl, vl, self.cos, self.sin, kv_len, self.scale)
            if fused:
                ops.linear(a, W.wo[i], resid=x, out=x, ss_out=ss)
                act = ops.mlp_act(x, W.wgu[i], ln=W.ln2[i], eps=W.eps, ss_in=ss)
                ops.linear(act, W.wd[i], resid=x, out=x, ss_out=ss)
            else:
                o = ops.linear(a, W.wo[i])
                h = ops.rmsnorm(o, W.ln2[i], W.eps, residual=x, sum_out=x)
                act = ops.mlp_act(h, W.wgu[i])
                d = ops.linear(act, W.wd[i])
        if fused:
            logits = ops.linear(x, W.lm_head, out_f32=True, ln=W.norm, eps=W.eps, ss_in=ss).unsqueeze(0)
        else:
            h = ops.rmsnorm(d, W.norm, W.eps, residual=x, sum_out=x)
            logits = ops.linear(h, W.lm_head, out_f32=True).unsqueeze(0)
        out = CausalLMOutput(logits)
        if probs is not None:
            from ..utils.sampling import norm_logits
            out.probs = norm_logits(logits[0, -1:], temperature=probs[0], top_k=-1, top_p=probs[1])[0]
        return out
