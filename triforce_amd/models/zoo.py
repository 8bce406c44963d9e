"""Shapes of the models the TriForce entry points name (the reference hard-codes hub ids:
test/on_chip.py:48-56, test/offloading_TP.py:56-67); offline they are built from these configs with
random-init weights, or loaded from a local HF directory.  Values are the public model cards'."""
from .config_yarn import LlamaConfig

_YARN_128K = dict(type="yarn", factor=32.0, original_max_position_embeddings=4096)

CONFIGS = {
    # NousResearch/Yarn-Llama-2-7b-128k
    "llama-7B-128K": dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                          max_position_embeddings=131072, rms_norm_eps=1e-5, rope_scaling=_YARN_128K,
                          _name_or_path="NousResearch/Yarn-Llama-2-7b-128k"),
    # NousResearch/Yarn-Llama-2-13b-128k
    "llama-13B-128K": dict(hidden_size=5120, intermediate_size=13824, num_hidden_layers=40, num_attention_heads=40,
                           max_position_embeddings=131072, rms_norm_eps=1e-5, rope_scaling=_YARN_128K,
                           _name_or_path="NousResearch/Yarn-Llama-2-13b-128k"),
    # LargeWorldModel/LWM-Text-Chat-128K: Llama-2-7B with plain RoPE, theta 1e7
    "lwm-128K": dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                     max_position_embeddings=131072, rms_norm_eps=1e-5, rope_theta=1e7,
                     _name_or_path="LargeWorldModel/LWM-Text-Chat-128K"),
    "lwm-128K-base": dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                          max_position_embeddings=131072, rms_norm_eps=1e-5, rope_theta=1e7,
                          _name_or_path="LargeWorldModel/LWM-Text-128K"),
    # JackFram/llama-68m
    "llama-68M": dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=2, num_attention_heads=12,
                      max_position_embeddings=2048, rms_norm_eps=1e-6, _name_or_path="JackFram/llama-68m"),
    # 2-layer D=128 YaRN toy used by BASELINE configs[0] ("tiny target") and the smoke paths
    "tiny": dict(hidden_size=256, intermediate_size=768, num_hidden_layers=2, num_attention_heads=2,
                 max_position_embeddings=131072, rms_norm_eps=1e-6,
                 rope_scaling=dict(type="yarn", factor=16.0, original_max_position_embeddings=256),
                 _name_or_path="tiny-yarn-target"),
}


def config(name):
    if name not in CONFIGS:
        raise NotImplementedError(f"unknown model {name!r}; known: {sorted(CONFIGS)}")
    return LlamaConfig.from_dict(CONFIGS[name])


def find_checkpoint(name):
    """Local directory holding the real checkpoint of a named model, or None (there is no hub access offline).
    Looked up, in order: $TRIFORCE_CKPT_DIR/<repo basename>, the HF cache ($HF_HOME/hub, ~/.cache/huggingface/hub:
    models--<org>--<repo>/snapshots/*), /root/models, /models, /data/models."""
    import glob
    import os
    repo = CONFIGS[name]["_name_or_path"]
    base = repo.split("/")[-1]
    cands = []
    if os.environ.get("TRIFORCE_CKPT_DIR"):
        cands += [os.path.join(os.environ["TRIFORCE_CKPT_DIR"], base), os.path.join(os.environ["TRIFORCE_CKPT_DIR"], repo)]
    hubs = [os.path.join(os.environ["HF_HOME"], "hub")] if os.environ.get("HF_HOME") else []
    hubs.append(os.path.expanduser("~/.cache/huggingface/hub"))
    for hub in hubs:
        cands += sorted(glob.glob(os.path.join(hub, "models--" + repo.replace("/", "--"), "snapshots", "*")))
    for root in ("/root/models", "/models", "/data/models"):
        cands += [os.path.join(root, base), os.path.join(root, repo)]
    for c in cands:
        if os.path.isfile(os.path.join(c, "config.json")) and \
                (glob.glob(os.path.join(c, "*.safetensors")) or glob.glob(os.path.join(c, "*.bin"))):
            return c
    return None
