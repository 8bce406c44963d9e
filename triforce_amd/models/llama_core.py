"""Weights, RoPE tables and the per-layer compute of a Llama decoder on the HIP ops.

Shared by the target model (modeling_llama.py), the 68M draft (modeling_llama_68m.py) and the
tensor-parallel engine (TP_llama.py).  q/k/v and gate/up projection weights are stored fused
(one skinny GEMM each instead of three / two): a decode step is bound by streaming the weights
once, so fewer, larger GEMMs and fused element-wise kernels are what matters on MI355X.
"""
import json
import math
import os
import zlib

import torch

from .. import ops
from .config_yarn import LlamaConfig


# ---- RoPE tables (same arithmetic as the reference, fp32 on the host, stored fp16) ----------
def rope_tables_plain(dim, max_pos, base):
    """reference models/modeling_llama.py:21-41."""
    inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
    freqs = torch.outer(torch.arange(max_pos, dtype=inv_freq.dtype), inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(torch.float16), emb.sin().to(torch.float16)


def rope_tables_yarn(dim, max_pos, factor, orig_max_pos, base=10000.0, beta_fast=32, beta_slow=1):
    """reference models/modeling_llama.py:50-124 (YaRN; base hard-wired to 10000 at :193)."""
    def corr_dim(n_rot):
        return (dim * math.log(orig_max_pos / (n_rot * 2 * math.pi))) / (2 * math.log(base))

    pos_freqs = base ** (torch.arange(0, dim, 2).float() / dim)
    inv_extrapolation = 1.0 / pos_freqs
    inv_interpolation = 1.0 / (factor * pos_freqs)
    low = float(max(math.floor(corr_dim(beta_fast)), 0))
    high = float(min(math.ceil(corr_dim(beta_slow)), dim - 1))
    if low == high:
        high += 0.001
    ramp = torch.clamp((torch.arange(dim // 2, dtype=torch.float32) - low) / (high - low), 0, 1)
    mask = 1 - ramp
    inv_freq = inv_interpolation * (1 - mask) + inv_extrapolation * mask
    mscale = 1.0 if factor <= 1 else 0.1 * math.log(factor) + 1.0
    freqs = torch.outer(torch.arange(max_pos, dtype=torch.float32), inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return (emb.cos() * mscale).to(torch.float16), (emb.sin() * mscale).to(torch.float16)


def rope_tables_for(cfg, force_plain=False):
    D = cfg.hidden_size // cfg.num_attention_heads
    rs = cfg.rope_scaling
    if rs is None or force_plain:
        return rope_tables_plain(D, cfg.max_position_embeddings, cfg.rope_theta)
    return rope_tables_yarn(D, cfg.max_position_embeddings, rs["factor"], rs["original_max_position_embeddings"])


def softmax_scale_for(head_dim):
    """fp16(1)/sqrt(fp16(D)) as the reference computes it (modeling_llama.py:240): 0.08837890625 for D=128."""
    return float(1 / torch.sqrt(torch.tensor(head_dim, dtype=torch.float16)))


class CausalLMOutput:
    __slots__ = ("logits", "probs")

    def __init__(self, logits, probs=None):
        self.logits = logits
        self.probs = probs


class LlamaWeights:
    """Device-resident fp16 weights with fused qkv / gate-up matrices.

    rank / world_size select a Megatron-style shard (TP_layers.py:126-147): q/k/v/gate/up rows,
    o/down columns; embed, norms and lm_head are replicated (TP_llama.py:91-94).
    """

    def __init__(self, cfg: LlamaConfig, device, rank=0, world_size=1):
        self.cfg, self.device, self.rank, self.world_size = cfg, torch.device(device), rank, world_size
        self.L = cfg.num_hidden_layers
        self.H = cfg.num_attention_heads
        self.D = cfg.hidden_size // self.H
        assert self.H % world_size == 0 and cfg.intermediate_size % world_size == 0
        self.H_local = self.H // world_size
        self.I_local = cfg.intermediate_size // world_size
        self.eps = cfg.rms_norm_eps
        self.embed = self.lm_head = self.norm = None
        self.wqkv, self.wo, self.wgu, self.wd, self.ln1, self.ln2 = [], [], [], [], [], []
        self.aligned = None       # models/aligned.py: spec + read-out state of an aligned synthetic pair
        self.capture = None       # list -> the forward appends its final residual stream (pre-norm), for calibration

    # -- loading ---------------------------------------------------------------------------
    def _shard_rows(self, w, n_local):
        return w[self.rank * n_local:(self.rank + 1) * n_local]

    def _shard_cols(self, w, n_local):
        return w[:, self.rank * n_local:(self.rank + 1) * n_local]

    def load_state_dict(self, sd):
        dev, hd = self.device, self.H_local * self.D

        def get(name):
            return sd[name].to(torch.float16)

        self.embed = get("model.embed_tokens.weight").to(dev)
        self.lm_head = get("lm_head.weight").to(dev) if "lm_head.weight" in sd else self.embed
        self.norm = get("model.norm.weight").to(dev)
        for i in range(self.L):
            p = f"model.layers.{i}."
            q = self._shard_rows(get(p + "self_attn.q_proj.weight"), hd)
            k = self._shard_rows(get(p + "self_attn.k_proj.weight"), hd)
            v = self._shard_rows(get(p + "self_attn.v_proj.weight"), hd)
            self.wqkv.append(torch.cat([q, k, v], dim=0).contiguous().to(dev))
            self.wo.append(self._shard_cols(get(p + "self_attn.o_proj.weight"), hd).contiguous().to(dev))
            g = self._shard_rows(get(p + "mlp.gate_proj.weight"), self.I_local)
            u = self._shard_rows(get(p + "mlp.up_proj.weight"), self.I_local)
            self.wgu.append(torch.cat([g, u], dim=0).contiguous().to(dev))
            self.wd.append(self._shard_cols(get(p + "mlp.down_proj.weight"), self.I_local).contiguous().to(dev))
            self.ln1.append(get(p + "input_layernorm.weight").to(dev))
            self.ln2.append(get(p + "post_attention_layernorm.weight").to(dev))
        return self.finalize()

    def init_random(self, seed, std=0.02):
        """Random-init directly on the device (no checkpoints offline): N(0,std) like
        modeling_llama.py:306-315; every rank draws the full matrix stream and keeps its shard
        only for the small configs — for big models each tensor is drawn shard-sized with a
        (seed, layer, name, rank)-derived generator."""
        dev, cfg = self.device, self.cfg
        hid, hd = cfg.hidden_size, self.H_local * self.D

        def draw(tag, *shape):
            g = torch.Generator(device=dev)
            g.manual_seed((seed * 1000003 + zlib.crc32(repr(tag).encode())) % (2 ** 31))
            return (torch.randn(*shape, generator=g, device=dev, dtype=torch.float32) * std).to(torch.float16)

        self.embed = draw("embed", cfg.vocab_size, hid)
        self.lm_head = draw("lm_head", cfg.vocab_size, hid)
        self.norm = torch.ones(hid, dtype=torch.float16, device=dev)
        for i in range(self.L):
            self.wqkv.append(draw(("qkv", i, self.rank), 3 * hd, hid))
            self.wo.append(draw(("o", i, self.rank), hid, hd))
            self.wgu.append(draw(("gu", i, self.rank), 2 * self.I_local, hid))
            self.wd.append(draw(("d", i, self.rank), hid, self.I_local))
            self.ln1.append(torch.ones(hid, dtype=torch.float16, device=dev))
            self.ln2.append(torch.ones(hid, dtype=torch.float16, device=dev))
        return self.finalize()

    def overwrite_random_(self, seed, std=0.02):
        """Re-draw every matrix IN PLACE with the values ``init_random(seed)`` would give (same storage, so captured
        hipGraphs and packed-weight pointers stay valid): lets one process measure two weight sets back to back."""
        dev, cfg = self.device, self.cfg
        assert isinstance(self.lm_head, ops.PackedLinear), "finalize() first"

        def draw_into(dst, tag):
            g = torch.Generator(device=dev)
            g.manual_seed((seed * 1000003 + zlib.crc32(repr(tag).encode())) % (2 ** 31))
            dst.copy_((torch.randn(*dst.shape, generator=g, device=dev, dtype=torch.float32) * std).to(torch.float16))

        tied = self.embed is self.lm_head.w
        draw_into(self.lm_head.w, "lm_head")
        self.lm_head.refresh_()
        if not tied:
            draw_into(self.embed, "embed")
        self.norm.fill_(1.0)
        for i in range(self.L):
            for lst, tag in ((self.wqkv, "qkv"), (self.wo, "o"), (self.wgu, "gu"), (self.wd, "d")):
                draw_into(lst[i].w, (tag, i, self.rank))
                lst[i].refresh_()
            self.ln1[i].fill_(1.0)
            self.ln2[i].fill_(1.0)
        self.aligned = None
        return self

    def init_aligned(self, spec, role, attn_keys=4096):
        """Aligned synthetic weights (models/aligned.py): real shapes, dense values, planted successor table so that
        draft / retrieval / full-cache forwards agree to a tunable degree."""
        from . import aligned
        return aligned.init_weights(self, spec, role, attn_keys=attn_keys)

    def finalize(self):
        """Wrap the GEMM weights: on a HIP device each gets its MFMA-packed copy for the decode kernel."""
        PL = ops.PackedLinear
        if isinstance(self.lm_head, PL):
            return self
        tied = self.lm_head is self.embed
        self.lm_head = PL(self.lm_head)
        if tied:
            self.embed = self.lm_head.w
        self.wqkv = [PL(w, rope=(self.H_local, self.D)) for w in self.wqkv]
        self.wo = [PL(w) for w in self.wo]
        self.wgu = [PL(w, split=2) for w in self.wgu]
        self.wd = [PL(w) for w in self.wd]
        return self

    def nbytes(self):
        ts = [self.embed, self.lm_head, self.norm] + self.wqkv + self.wo + self.wgu + self.wd + self.ln1 + self.ln2
        ts = [ops._w(t) for t in ts]
        return sum(t.numel() * t.element_size() for t in ts)


def load_checkpoint_state_dict(path):
    """HF directory with *.safetensors (or pytorch_model*.bin) -> flat state dict on CPU."""
    sd = {}
    st_files = sorted(f for f in os.listdir(path) if f.endswith(".safetensors"))
    if st_files:
        from safetensors.torch import load_file
        for f in st_files:
            sd.update(load_file(os.path.join(path, f)))
        return sd
    bins = sorted(f for f in os.listdir(path) if f.endswith(".bin"))
    if not bins:
        raise FileNotFoundError(f"no *.safetensors / *.bin weights under {path}")
    for f in bins:
        sd.update(torch.load(os.path.join(path, f), map_location="cpu"))
    return sd


def parse_random_spec(name_or_path):
    """``random:<seed>`` -> seed (int) else None."""
    if isinstance(name_or_path, str) and name_or_path.startswith("random"):
        parts = name_or_path.split(":")
        return int(parts[1]) if len(parts) > 1 and parts[1] else 0
    return None


def save_config(cfg, path):
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump({k: v for k, v in cfg.to_dict().items() if not k.startswith("_")}, f, indent=1)
