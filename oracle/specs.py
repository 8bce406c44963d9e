"""Seeded tiny-model specs shared by the golden generator and the tests.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  No checkpoints exist offline, so every
parity case is a (config, seed) pair; weights are regenerated deterministically from the
torch CPU generator on both sides (the GPU box runs the same image, hence the same torch).

Config keys follow the HF Llama ``config.json`` the reference loads
(models/config_yarn.py:31; consumed at modeling_llama.py:166-198).
"""
import torch


def llama_config(hidden_size, intermediate_size, num_hidden_layers, num_attention_heads,
                 vocab_size=32000, max_position_embeddings=4096, rope_theta=10000.0,
                 rope_scaling=None, rms_norm_eps=1e-6, name="tiny"):
    return dict(
        hidden_size=hidden_size, intermediate_size=intermediate_size,
        num_hidden_layers=num_hidden_layers, num_attention_heads=num_attention_heads,
        num_key_value_heads=num_attention_heads, vocab_size=vocab_size,
        max_position_embeddings=max_position_embeddings, rope_theta=rope_theta,
        rope_scaling=rope_scaling, rms_norm_eps=rms_norm_eps, hidden_act="silu",
        _name_or_path=name,
    )


# --- the shapes used across tests/golden --------------------------------------------------
def draft_68m_config(vocab_size=32000):
    """JackFram/llama-68m shape [model-card]: 2 layers, hidden 768, 12 heads (D=64), I=3072."""
    return llama_config(768, 3072, 2, 12, vocab_size=vocab_size, max_position_embeddings=2048,
                        name="llama-68m-shaped")


def tiny_target_config(vocab_size=32000, layers=2, hidden=256, heads=2, max_pos=4096):
    """Tiny YaRN target with D=128 heads (same head_dim as Llama-2-7B), factor 16 / orig 256."""
    return llama_config(hidden, hidden * 3, layers, heads, vocab_size=vocab_size,
                        max_position_embeddings=max_pos,
                        rope_scaling=dict(type="yarn", factor=16.0, original_max_position_embeddings=256),
                        name="tiny-yarn-target")


def llama2_7b_128k_config():
    """NousResearch/Yarn-Llama-2-7b-128k [model-card]."""
    return llama_config(4096, 11008, 32, 32, max_position_embeddings=131072, rms_norm_eps=1e-5,
                        rope_scaling=dict(type="yarn", factor=32.0, original_max_position_embeddings=4096),
                        name="Yarn-Llama-2-7b-128k(random-init)")


def llama2_13b_128k_config():
    """NousResearch/Yarn-Llama-2-13b-128k [model-card]."""
    return llama_config(5120, 13824, 40, 40, max_position_embeddings=131072, rms_norm_eps=1e-5,
                        rope_scaling=dict(type="yarn", factor=32.0, original_max_position_embeddings=4096),
                        name="Yarn-Llama-2-13b-128k(random-init)")


def lwm_text_chat_128k_config():
    """LargeWorldModel/LWM-Text-Chat-128K [model-card]: Llama-2-7B with plain RoPE, theta 1e7."""
    return llama_config(4096, 11008, 32, 32, max_position_embeddings=131072, rope_theta=1e7,
                        rms_norm_eps=1e-5, name="LWM-Text-Chat-128K(random-init)")


def random_state_dict(cfg, seed, std=0.02, head_std=None, dtype=torch.float16, device="cpu"):
    """HF-keyed state dict, N(0,std) linears/embedding (modeling_llama.py:306-315 init),
    norm weights 1 + 0.1*N(0,1) so the weight-multiply order in RMSNorm is actually exercised."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    H, I, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]

    def w(*shape, s=std):
        return (torch.randn(*shape, generator=g, dtype=torch.float32) * s).to(dtype).to(device)

    def nw():
        return (1.0 + 0.1 * torch.randn(H, generator=g, dtype=torch.float32)).to(dtype).to(device)

    sd = {"model.embed_tokens.weight": w(V, H)}
    for i in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{i}."
        sd[p + "self_attn.q_proj.weight"] = w(H, H)
        sd[p + "self_attn.k_proj.weight"] = w(H, H)
        sd[p + "self_attn.v_proj.weight"] = w(H, H)
        sd[p + "self_attn.o_proj.weight"] = w(H, H)
        sd[p + "mlp.gate_proj.weight"] = w(I, H)
        sd[p + "mlp.up_proj.weight"] = w(I, H)
        sd[p + "mlp.down_proj.weight"] = w(H, I)
        sd[p + "input_layernorm.weight"] = nw()
        sd[p + "post_attention_layernorm.weight"] = nw()
    sd["model.norm.weight"] = nw()
    sd["lm_head.weight"] = w(V, H, s=head_std if head_std is not None else std)
    return sd


def shard_of(cfg, sd, rank, world):
    """(config, state dict) of ONE tensor-parallel rank's network, sliced as the reference slices a layer
    (models/TP_layers.py:126-147: q/k/v/gate/up split on dim 0, o/down on dim 1, `.split(n // world)[rank]`; embedding,
    norms and lm_head replicated, TP_llama.py:91-94).  With the all-reduces taken out (a one-process group) a rank
    computes exactly this narrower network — hidden size unchanged, H / world heads of the model's head_dim, I / world
    MLP columns — which is what the device's shard forward is compared with at the 13B / TP = 8 widths."""
    H, I, hid = cfg["num_attention_heads"], cfg["intermediate_size"], cfg["hidden_size"]
    assert H % world == 0 and I % world == 0
    D = hid // H
    scfg = dict(cfg, num_attention_heads=H // world, num_key_value_heads=cfg["num_key_value_heads"] // world,
                intermediate_size=I // world, head_dim=D, _name_or_path=f"{cfg['_name_or_path']}[rank {rank}/{world}]")
    out = {}
    for name, w in sd.items():
        if any(k in name for k in ("q_proj", "k_proj", "v_proj")):
            w = w.split((H * D) // world, dim=0)[rank]
        elif "o_proj" in name:
            w = w.split(hid // world, dim=1)[rank]
        elif "gate_proj" in name or "up_proj" in name:
            w = w.split(I // world, dim=0)[rank]
        elif "down_proj" in name:
            w = w.split(I // world, dim=1)[rank]
        out[name] = w.contiguous()
    return scfg, out


def tensor_digest(t):
    """sha256 of a tensor's contiguous bytes (fixtures that pin large seeded tensors without storing them)."""
    import hashlib
    return hashlib.sha256(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()


def random_prompt(vocab_size, length, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randint(3, vocab_size, (1, length), generator=g, dtype=torch.long)
