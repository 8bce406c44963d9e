"""Llama config schema for the TriForce path (role of models/config_yarn.py:31-193 in the reference).

A plain attribute bag that reads the HF ``config.json`` keys the path consumes
(modeling_llama.py:166-198): no transformers dependency.
"""
import json
import os

_DEFAULTS = dict(
    vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
    num_attention_heads=32, num_key_value_heads=None, hidden_act="silu", max_position_embeddings=2048,
    rms_norm_eps=1e-6, rope_theta=10000.0, rope_scaling=None, attention_bias=False, pad_token_id=None,
    bos_token_id=1, eos_token_id=2, _name_or_path="",
)


class LlamaConfig:
    def __init__(self, **kw):
        for k, v in _DEFAULTS.items():
            setattr(self, k, kw.pop(k, v))
        for k, v in kw.items():
            setattr(self, k, v)
        if self.num_key_value_heads is None:
            self.num_key_value_heads = self.num_attention_heads
        self._validate()

    def _validate(self):
        # role of config_yarn.py:170-193 (_rope_scaling_validation), reduced to what the path supports
        if self.hidden_act != "silu":
            raise ValueError(f"hidden_act {self.hidden_act!r} is not supported (silu only)")
        if self.attention_bias:
            raise ValueError("attention_bias is not supported")
        if self.num_key_value_heads != self.num_attention_heads:
            # the reference's retrieval cache is MHA-only
            raise ValueError("GQA is not supported: num_key_value_heads must equal num_attention_heads")
        rs = self.rope_scaling
        if rs is not None:
            if not isinstance(rs, dict) or "type" not in rs or "factor" not in rs:
                raise ValueError(f"`rope_scaling` must be a dict with `type` and `factor`, got {rs}")
            if rs["type"] != "yarn":
                raise ValueError(f"Unknown RoPE scaling type {rs['type']}")
            if float(rs["factor"]) <= 1.0:
                raise ValueError(f"`rope_scaling`'s factor must be > 1, got {rs['factor']}")
            if "original_max_position_embeddings" not in rs:
                raise ValueError("yarn rope_scaling needs original_max_position_embeddings")

    @property
    def head_dim(self):
        return self.hidden_size // self.num_attention_heads

    @classmethod
    def from_dict(cls, d):
        return cls(**dict(d))

    @classmethod
    def from_pretrained(cls, path):
        with open(os.path.join(path, "config.json")) as f:
            d = json.load(f)
        d = {k: v for k, v in d.items() if k in _DEFAULTS or k.startswith("rope")}
        d["_name_or_path"] = path
        return cls(**d)

    def to_dict(self):
        return {k: getattr(self, k) for k in _DEFAULTS}
