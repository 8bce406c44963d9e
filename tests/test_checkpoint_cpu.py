"""The checkpoint path the entry points drop into (reference test/on_chip.py:48-56 `LlamaForCausalLM.from_pretrained(hub id)`,
test/offloading_TP.py:97-102 rank-by-rank `init_parameters(hf_model)`): no hub or trained weights exist offline, so a seeded
tiny model is written the way SURVEY 8c pitfall (ii) prescribes — `config.json` (with the MODEL-CARD `rope_scaling` dict,
`finetuned` and the 5.x `rope_type` key included) + `*.safetensors` — and loaded back through every loader the scripts use.
CPU: tensors, RoPE tables and config must be bit-identical to the `from_state_dict` construction; world 2 over gloo:
`cli.shard_weights` must leave each rank with exactly the reference's head / column shard (oracle.specs.shard_of, pinned to
TP_layers.py:126-147 by tests/golden/tp_shards.json).  The GPU half (logits, the script itself) is tests/test_gpu_configs.py."""
import json
import os
import sys
import traceback

import pytest
import torch
import torch.distributed as dist

from oracle import specs
from tests.test_tp_cpu import _run_world

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MODEL_CARD_ROPE = {"type": "yarn", "factor": 16.0, "original_max_position_embeddings": 256, "finetuned": True,
                   "rope_type": "yarn"}            # (`finetuned`: Yarn-Llama-2 cards; `rope_type`: written by transformers 5.x)


def write_checkpoint(path, cfg_dict, seed, shards=1, head_std=0.05):
    """config.json + model[-0000i-of-0000n].safetensors of a seeded tiny model, HF key names; returns the state dict."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    sd = specs.random_state_dict(cfg_dict, seed, head_std=head_std)
    hf = dict(cfg_dict, architectures=["LlamaForCausalLM"], model_type="llama", torch_dtype="float16",
              transformers_version="4.37.2", tie_word_embeddings=False, use_cache=True)
    hf.pop("_name_or_path", None)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(hf, f, indent=1)
    keys = sorted(sd)
    for i in range(shards):
        part = {k: sd[k].contiguous() for k in keys[i::shards]}
        name = "model.safetensors" if shards == 1 else f"model-{i + 1:05d}-of-{shards:05d}.safetensors"
        save_file(part, os.path.join(path, name))
    return sd


def tiny_cfgs():
    t = specs.tiny_target_config(vocab_size=512, layers=2, hidden=256, heads=2)
    t["rope_scaling"] = dict(MODEL_CARD_ROPE)
    d = specs.llama_config(128, 256, 2, 2, vocab_size=512, max_position_embeddings=2048, name="tiny-draft")
    return t, d


def _tensors(W):
    from triforce_amd import ops
    out = {"embed": W.embed, "lm_head": ops._w(W.lm_head), "norm": W.norm}
    for i in range(len(W.wqkv)):
        for n in ("wqkv", "wo", "wgu", "wd", "ln1", "ln2"):
            out[f"{n}{i}"] = ops._w(getattr(W, n)[i])
    return out


@pytest.mark.parametrize("shards", [1, 3])
def test_from_pretrained_equals_from_state_dict(tmp_path, shards):
    from triforce_amd.models.config_yarn import LlamaConfig
    from triforce_amd.models.llama_core import rope_tables_for
    from triforce_amd.models.modeling_llama import LlamaForCausalLM
    from triforce_amd.models.modeling_llama_68m import LlamaForCausalLM as Draft
    tcfg, dcfg = tiny_cfgs()
    for cls, cfg, seed, sub in ((LlamaForCausalLM, tcfg, 11, "target"), (Draft, dcfg, 12, "draft")):
        path = str(tmp_path / sub)
        sd = write_checkpoint(path, cfg, seed, shards=shards)
        loaded = cls.from_pretrained(path, torch_dtype=torch.float16, device_map="cpu")
        built = cls.from_state_dict(LlamaConfig.from_dict(cfg), sd, "cpu")
        a, b = _tensors(loaded.weights), _tensors(built.weights)
        assert a.keys() == b.keys()
        for k in a:
            assert a[k].dtype == torch.float16 and torch.equal(a[k], b[k]), k
        for k, v in cfg.items():
            if k == "_name_or_path":
                assert loaded.config._name_or_path == path                  # what decoding.py:31 prints
            elif k == "rope_scaling" and v is not None:
                got = loaded.config.rope_scaling
                assert got["type"] == "yarn" and got["factor"] == v["factor"] and \
                    got["original_max_position_embeddings"] == v["original_max_position_embeddings"]
            else:
                assert getattr(loaded.config, k) == v, k
        for x, y in zip(rope_tables_for(loaded.config), rope_tables_for(built.config)):
            assert torch.equal(x, y)


def test_missing_weights_and_bad_rope_scaling_fail_loudly(tmp_path):
    from triforce_amd.models.config_yarn import LlamaConfig
    from triforce_amd.models.modeling_llama import LlamaForCausalLM
    tcfg, _ = tiny_cfgs()
    path = str(tmp_path / "empty")
    os.makedirs(path)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump({k: v for k, v in tcfg.items() if k != "_name_or_path"}, f)
    with pytest.raises(FileNotFoundError):
        LlamaForCausalLM.from_pretrained(path, device_map="cpu")
    bad = dict(tcfg, rope_scaling={"type": "linear", "factor": 4.0})
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump({k: v for k, v in bad.items() if k != "_name_or_path"}, f)
    with pytest.raises(ValueError):
        LlamaConfig.from_pretrained(path)


def test_zoo_finds_a_checkpoint_under_TRIFORCE_CKPT_DIR(tmp_path, monkeypatch):
    from triforce_amd.models import zoo
    name = "llama-68M"
    repo = zoo.CONFIGS[name]["_name_or_path"]
    monkeypatch.setenv("TRIFORCE_CKPT_DIR", str(tmp_path))
    monkeypatch.setenv("HF_HOME", str(tmp_path / "no-hf-home"))
    assert zoo.find_checkpoint(name) in (None,) or not zoo.find_checkpoint(name).startswith(str(tmp_path))
    _, dcfg = tiny_cfgs()
    write_checkpoint(str(tmp_path / repo.split("/")[-1]), dcfg, 5)
    assert zoo.find_checkpoint(name) == str(tmp_path / repo.split("/")[-1])


def _shard_worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        torch.set_num_threads(2)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from triforce_amd.models.config_yarn import LlamaConfig
        from triforce_amd.models.TP_llama import DistributedLlama
        from triforce_amd.utils import cli
        path = os.environ["TF_TEST_CKPT"]
        llm = DistributedLlama(path, device="cpu", local_rank=rank, world_size=world, prefill=64, gen_len=16,
                               retrieval_budget=32, kv_offload=True, on_chip_layers=2, gamma=2)
        assert isinstance(llm.config.model_config, LlamaConfig) and llm.config.model_config._name_or_path == path
        cli.shard_weights(llm, path, rank, world)                     # offloading_TP.py:97-102
        W = llm.weights
        got = {k: v.contiguous().view(torch.int16).numpy().copy() for k, v in _tensors(W).items()}   # by value (bit patterns)
        q.put((rank, "ok", got, W.H_local, W.I_local))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, "error", traceback.format_exc()))


def test_shard_weights_from_a_checkpoint_at_world_2(tmp_path, monkeypatch):
    tcfg, _ = tiny_cfgs()
    path = str(tmp_path / "target")
    sd = write_checkpoint(path, tcfg, 21, shards=2)
    monkeypatch.setenv("TF_TEST_CKPT", path)
    world = 2
    outs = _run_world(_shard_worker, world)
    H, D, I = tcfg["num_attention_heads"], tcfg["hidden_size"] // tcfg["num_attention_heads"], tcfg["intermediate_size"]
    for rank in range(world):
        _, _, got, Hl, Il = outs[rank]
        got = {k: torch.from_numpy(v).view(torch.float16) for k, v in got.items()}
        assert (Hl, Il) == (H // world, I // world)
        _, want = specs.shard_of(tcfg, sd, rank, world)                # the reference's slicing (TP_layers.py:126-147)
        for i in range(tcfg["num_hidden_layers"]):
            q, k, v = (want[f"model.layers.{i}.self_attn.{n}_proj.weight"] for n in "qkv")
            assert torch.equal(got[f"wqkv{i}"], torch.cat([q, k, v], 0)), (rank, i, "wqkv")
            assert torch.equal(got[f"wo{i}"], want[f"model.layers.{i}.self_attn.o_proj.weight"]), (rank, i, "wo")
            gu = torch.cat([want[f"model.layers.{i}.mlp.gate_proj.weight"], want[f"model.layers.{i}.mlp.up_proj.weight"]], 0)
            assert torch.equal(got[f"wgu{i}"], gu), (rank, i, "wgu")
            assert torch.equal(got[f"wd{i}"], want[f"model.layers.{i}.mlp.down_proj.weight"]), (rank, i, "wd")
        assert torch.equal(got["embed"], sd["model.embed_tokens.weight"]) and torch.equal(got["lm_head"], sd["lm_head.weight"])
