"""Tensor-parallel sharding descriptors (role of the reference's models/TP_layers.py).

The reference slices every HF layer on the host and pins the slices (DistributedLlamaLayer.init_parameters
:126-163): q/k/v/gate/up are split by rows (dim 0), o/down by columns (dim 1).  Here the slicing lives in
llama_core.LlamaWeights (rank / world_size), which also fuses q|k|v and gate|up of the local shard into
one matrix each; this module keeps the reference's config type and a per-layer view for callers that
want the individual matrices.
"""


class DistributedOffloadingConfig:
    """Model config + this rank's coordinates (reference TP_layers.py:5-16)."""

    def __init__(self, config, local_rank=0, world_size=1) -> None:
        self.model_config = config
        self.local_rank = local_rank
        self.world_size = world_size
        self.hidden_size = config.hidden_size
        self.intermediate_size = config.intermediate_size
        self.num_attention_heads = config.num_attention_heads
        self.num_key_value_heads = config.num_key_value_heads
        self.num_hidden_layers = config.num_hidden_layers
        self.vocab_size = config.vocab_size
        self.max_position_embeddings = config.max_position_embeddings
        self.rms_norm_eps = config.rms_norm_eps
        self._name_or_path = getattr(config, "_name_or_path", "")
        if self.num_attention_heads % world_size or self.intermediate_size % world_size:
            raise ValueError(f"heads ({self.num_attention_heads}) and intermediate size ({self.intermediate_size}) "
                             f"must be divisible by the TP degree ({world_size})")


class DistributedLlamaLayer:
    """Read-only view of one layer's local shard (names of reference TP_layers.py:100-163)."""

    def __init__(self, layer_idx, weights) -> None:
        from .. import ops
        W, hd, I = weights, weights.H_local * weights.D, weights.I_local
        self.layer_idx = layer_idx
        qkv, gu = ops._w(W.wqkv[layer_idx]), ops._w(W.wgu[layer_idx])
        self.wq, self.wk, self.wv = qkv[:hd], qkv[hd:2 * hd], qkv[2 * hd:]
        self.wo = ops._w(W.wo[layer_idx])
        self.gate_proj, self.up_proj = gu[:I], gu[I:]
        self.down_proj = ops._w(W.wd[layer_idx])
        self.input_layernorm_weight = W.ln1[layer_idx]
        self.post_attention_layernorm_weight = W.ln2[layer_idx]
        self.input_layernorm_variance_epsilon = self.post_attention_layernorm_variance_epsilon = W.eps
