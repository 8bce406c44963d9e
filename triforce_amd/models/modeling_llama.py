"""Target model (Llama-2-7B/13B-128K, LWM-Text-Chat-128K) on the HIP ops — host-side mirror of
the reference's models/modeling_llama.py: same call signature
``model(input_ids=, kv_cache=, graph_cache=, position_ids=, spec=).logits`` (:384-414), same
dispatch between full-cache forward, retrieval-cache (spec) forward and the q_len==1 retrieval
build (:226-238), same numerics.

A decode-sized forward (<= 32 rows) is 6 launches per layer, all hand-written: qkv GEMM (RMSNorm prologue,
RoPE + KV-append epilogue), split-KV MFMA attention + merge, o GEMM (+residual), gate|up GEMM (RMSNorm prologue,
SwiGLU epilogue), down GEMM (+residual).  Prefill chunks (128 rows) run hipBLASLt GEMMs with the stand-alone
RMSNorm / RoPE-append / SwiGLU kernels and the 128-row block attention.
"""
import torch

from .. import ops
from .cache import RetrievalCache
from .config_yarn import LlamaConfig
from .llama_core import (CausalLMOutput, LlamaWeights, load_checkpoint_state_dict, parse_random_spec,
                         rope_tables_for, softmax_scale_for)


class LlamaForCausalLM:
    def __init__(self, config: LlamaConfig, device="cuda:0"):
        self.config = config
        self.device = torch.device(device)
        self.dtype = torch.float16
        self.weights = LlamaWeights(config, self.device)
        cos, sin = rope_tables_for(config)
        self.cos, self.sin = cos.to(self.device), sin.to(self.device)
        self.scale = softmax_scale_for(config.hidden_size // config.num_attention_heads)
        self.vocab_size = config.vocab_size

    # -- construction ------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, name_or_path, torch_dtype=torch.float16, device_map="cuda:0", config=None, **_):
        """Local HF directory, or ``random:<seed>`` with an explicit ``config`` (no hub access offline)."""
        assert torch_dtype == torch.float16, "the TriForce path is fp16"
        seed = parse_random_spec(name_or_path)
        if seed is not None:
            assert config is not None, "random:<seed> needs config="
            return cls(config, device_map).init_random(seed)
        from .aligned import parse_spec
        spec = parse_spec(name_or_path)
        if spec is not None:                                # aligned[:draft_acc[:retrieval_acc[:seed]]]
            assert config is not None, "aligned:... needs config="
            return cls(config, device_map).init_aligned(spec, attn_keys=_.get("attn_keys", 4096))
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

    def init_aligned(self, spec, attn_keys=4096):
        """Aligned synthetic weights (models/aligned.py), this model in the target role."""
        self.weights.init_aligned(spec, "target", attn_keys=attn_keys)
        return self

    def eval(self):
        return self

    # -- forward -----------------------------------------------------------------------------
    @torch.inference_mode()
    def __call__(self, input_ids, kv_cache=None, graph_cache=None, position_ids=None, spec=False,
                 attention_mask=None, storage_ids=None, gamma_offset=0, rebuild_retrieval=False, dev_len=None):
        return self.forward(input_ids, kv_cache, graph_cache, position_ids, spec, rebuild_retrieval, dev_len)

    def forward(self, input_ids, kv_cache, graph_cache=None, position_ids=None, spec=False, rebuild_retrieval=False,
                dev_len=None, last_rows=None):
        """last_rows = k: logits of the trailing k rows only (chunked prefill: utils/graph_infer.py chunked_prefill).
        dev_len = (slot_dev, sk_dev) int32 device scalars: the hipGraph-capturable form of the full-cache decode
        forward — the append slot and the key count are read from device memory by the kernels (tf_skinny_qkv_rope
        slot0_dev, tf_attn_decode sk_dev), position_ids must be given, the launch is sized by the cache capacity and
        kv_cache.seq_len is NOT advanced (the caller does that after the replay)."""
        W = self.weights
        H, D = W.H, W.D
        q_len = input_ids.shape[1]
        if dev_len is not None:
            assert position_ids is not None and not spec and q_len <= ops.SKINNY_MAX_ROWS
        if position_ids is None:                        # reference modeling_llama.py:345-349
            position_ids = torch.arange(kv_cache.seq_len, kv_cache.seq_len + q_len, dtype=torch.long,
                                        device=self.device).unsqueeze(0)
        pos = position_ids.reshape(-1).contiguous()
        build = (not spec) and q_len == 1 and isinstance(graph_cache, RetrievalCache)
        # periodic rebuild (described in the reference's blog, absent from its code): during a
        # target verify, re-select the prefill chunks with the query of the first — already confirmed — token
        rebuild = (not spec) and q_len > 1 and rebuild_retrieval and isinstance(graph_cache, RetrievalCache)
        streaming = (not spec) and hasattr(kv_cache, "begin_forward")     # host-offloaded KV (test/offloading.py)
        if streaming:
            kv_cache.begin_forward()
        # decode-sized blocks run the fused kernels: [norm ->] qkv GEMM -> RoPE -> KV append in one launch, the
        # residual adds in the o / down GEMM epilogues, the post-attention norm in the gate|up GEMM prologue
        mode = ops.FUSE_MODE if (ops.can_fuse_rows(q_len, W.embed, W.wqkv[0], W.wo[0], W.wgu[0], W.wd[0], W.lm_head)
                                 and W.wqkv[0].wp_rope is not None) else "none"
        fused = mode in ("all", "all2")
        # the fused layer keeps residual stream / attention output / SwiGLU output k-octet-major (ops.Act): the GEMMs' B
        # operand is then read in 256-byte runs (ops.py, "activation layouts")
        packed = fused and ops.act_packed(q_len)
        x = ops.embed_rows(W.embed, input_ids, packed)   # (q, hidden) fp16 gather
        ss = ops.ss_buffer(x.shape[1], x.device) if mode == "all" else None     # sum(x^2) hand-off between GEMMs
        d = None
        for i in range(W.L):
            if spec:                                    # :226-227  retrieval-cache forward
                kl, vl = graph_cache.layer_kv(i)
                assert q_len == graph_cache.gamma + 1, "spec forward takes exactly gamma+1 tokens (cache.py:184-189)"
                slot, sk = graph_cache.spec_slot, graph_cache.real_budget
            elif dev_len is not None:                   # captured full-cache forward: lengths live on the device
                kl, vl = kv_cache.layer_kv(i)
                slot, sk = 0, kv_cache.max_budget
            else:                                       # :228-238  full-cache forward
                kl, vl = kv_cache.layer_kv(i)
                slot = kv_cache.append_slot(i, q_len)
                sk = slot + q_len
            if fused:
                q = ops.qkv_rope(x, W.wqkv[i], W.ln1[i], W.eps, self.cos, self.sin, pos, kl, vl, slot, H, D,
                                 ss_in=ss if i > 0 else None, slot0_dev=dev_len[0] if dev_len is not None else None)
            else:
                if d is None:
                    h = ops.rmsnorm(x, W.ln1[i], W.eps)
                else:                                   # x += mlp_out of the previous layer, fused into the norm
                    h = ops.rmsnorm(d, W.ln1[i], W.eps, residual=x, sum_out=x)
                if mode == "rope":
                    q = ops.qkv_rope(h, W.wqkv[i], None, 0.0, self.cos, self.sin, pos, kl, vl, slot, H, D)
                else:
                    q = ops.rope_append(ops.linear(h, W.wqkv[i]), self.cos, self.sin, pos, kl, vl, slot, H, D)
            if spec:
                a = ops.attn_decode(q, kl, vl, sk, self.scale, packed=packed)
            elif dev_len is not None:
                assert fused, "the captured full-cache forward needs the fused decode kernels"
                a = ops.attn_decode(q, kl, vl, sk, self.scale, sk_dev=dev_len[1], packed=packed)
            else:
                if build:
                    if not graph_cache.init_graph:
                        graph_cache.init_graph_cache(kv_cache, q, i)
                    else:
                        graph_cache.update_graph_cache_retrieval(kv_cache, q, i)
                elif rebuild:                           # generated tail is re-copied by update_graph_cache() after accept
                    graph_cache.init_graph_cache(kv_cache, q[:1], i)
                a = ops.attn_decode(q, kl, vl, sk, self.scale, packed=True) if packed else \
                    ops.attn_prefill(q, kl, vl, sk, self.scale)
                if streaming:
                    kv_cache.layer_done(i, slot, q_len)
            if fused:
                ops.linear(a, W.wo[i], resid=x, out=x, ss_out=ss)                               # x += attn_out
                act = ops.mlp_act(x, W.wgu[i], ln=W.ln2[i], eps=W.eps, ss_in=ss)
                ops.linear(act, W.wd[i], resid=x, out=x, ss_out=ss)                             # x += mlp_out
            else:
                o = ops.linear(a, W.wo[i])
                h = ops.rmsnorm(o, W.ln2[i], W.eps, residual=x, sum_out=x)           # x += attn_out
                act = ops.mlp_act(h, W.wgu[i])
                d = ops.linear(act, W.wd[i])
        if streaming:
            kv_cache.end_forward()
        if fused:
            if W.capture is not None:
                W.capture.append(x.rows() if packed else x.clone())
            logits = ops.linear(x, W.lm_head, out_f32=True, ln=W.norm, eps=W.eps, ss_in=ss).unsqueeze(0)
        else:
            h = ops.rmsnorm(d, W.norm, W.eps, residual=x, sum_out=x)
            if W.capture is not None:
                W.capture.append(x.clone())
            if last_rows is not None and last_rows < h.shape[0]:
                h = h[-last_rows:]
            logits = ops.linear(h, W.lm_head, out_f32=True).unsqueeze(0)           # (1, q, V) fp32  (:408-409)
        return CausalLMOutput(logits)
