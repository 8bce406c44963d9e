"""CPU counterparts of tests/test_gpu_configs.py: the scaffolding of the shard-width / plain-RoPE / fp64-truth parity
tests checked where no GPU is needed — oracle.specs.shard_of against the product's own slicing and against the
reference's rule, the one-rank-of-W engine (host logic over the oracle ops, one-process gloo group) against the oracle's
shard network, and the fp64-accumulating restatement against the oracle.  No kernel is validated here."""
import math
import os
import socket

import pytest
import torch
import torch.distributed as dist

from oracle import ref_model as M
from oracle import ref_ops as R
from oracle import specs


def _cfg(hidden=512, inter=1024, heads=8):
    return specs.llama_config(hidden, inter, 1, heads, rms_norm_eps=1e-5, max_position_embeddings=4096,
                              rope_scaling=dict(type="yarn", factor=16.0, original_max_position_embeddings=256))


def test_shard_of_follows_the_reference_split_rule_and_the_products_slicing():
    """TP_layers.py:126-147: q/k/v/gate/up `.split(n // world, dim=0)[rank]`, o/down `.split(n // world, dim=1)[rank]`."""
    from triforce_amd.models.config_yarn import LlamaConfig
    from triforce_amd.models.llama_core import LlamaWeights
    cfg = _cfg()
    sd = specs.random_state_dict(cfg, 3)
    rank, world = 3, 4
    scfg, ssd = specs.shard_of(cfg, sd, rank, world)
    assert (scfg["num_attention_heads"], scfg["num_key_value_heads"], scfg["intermediate_size"], scfg["head_dim"]) == (2, 2, 256, 64)
    assert R.head_dim_of(scfg) == 64 and R.head_dim_of(cfg) == 64
    p = "model.layers.0."
    hd = 2 * 64
    assert torch.equal(ssd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.weight"][rank * hd:(rank + 1) * hd])
    assert torch.equal(ssd[p + "self_attn.o_proj.weight"], sd[p + "self_attn.o_proj.weight"][:, rank * hd:(rank + 1) * hd])
    assert torch.equal(ssd[p + "mlp.down_proj.weight"], sd[p + "mlp.down_proj.weight"][:, rank * 256:(rank + 1) * 256])
    assert torch.equal(ssd["lm_head.weight"], sd["lm_head.weight"])
    W = LlamaWeights(LlamaConfig.from_dict(cfg), "cpu", rank=rank, world_size=world)
    W.load_state_dict(sd)

    def dense(w):
        return w.w if hasattr(w, "w") else w
    qkv = torch.cat([ssd[p + f"self_attn.{n}_proj.weight"] for n in "qkv"], dim=0)
    assert torch.equal(dense(W.wqkv[0]).cpu(), qkv)
    assert torch.equal(dense(W.wo[0]).cpu(), ssd[p + "self_attn.o_proj.weight"])
    assert torch.equal(dense(W.wgu[0]).cpu(), torch.cat([ssd[p + "mlp.gate_proj.weight"], ssd[p + "mlp.up_proj.weight"]], dim=0))
    assert torch.equal(dense(W.wd[0]).cpu(), ssd[p + "mlp.down_proj.weight"])


def _one_process_gloo():
    if not dist.is_initialized():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        dist.init_process_group("gloo", rank=0, world_size=1, init_method=f"tcp://127.0.0.1:{port}")


@pytest.mark.parametrize("gamma", [6, 16])
def test_one_rank_of_a_tp_engine_computes_the_oracles_shard_network(cpu_ops, gamma):
    """What tests/test_gpu_configs.py::test_13b_tp8_shard_layer_logits_match_oracle relies on: with a one-process
    group a DistributedLlama built as rank r of W computes exactly the network shard_of() describes."""
    from triforce_amd.models.config_yarn import LlamaConfig
    from triforce_amd.models.TP_llama import DistributedLlama
    _one_process_gloo()
    cfg = _cfg()
    sd = specs.random_state_dict(cfg, 4)
    rank, world, budget, prefill = 3, 4, 64, 256
    scfg, ssd = specs.shard_of(cfg, sd, rank, world)
    llm = DistributedLlama("unused", config=LlamaConfig.from_dict(cfg), device="cpu", local_rank=rank, world_size=world,
                           prefill=prefill, gen_len=32, retrieval_budget=budget, retrieval_chunk_size=8, kv_offload=True,
                           on_chip_layers=1, gamma=gamma)
    llm.init_parameters(sd)
    ot = M.OracleTarget(scfg, ssd)
    ogc = M.RetrievalCacheO(scfg, budget, prefill, 8, gamma)
    gen = torch.Generator().manual_seed(8)
    ogc.key_cache.copy_(torch.randn(ogc.key_cache.shape, generator=gen).half())
    ogc.value_cache.copy_(torch.randn(ogc.value_cache.shape, generator=gen).half())
    llm.retrieval_cache.k.copy_(ogc.key_cache.permute(0, 2, 1, 3))
    llm.retrieval_cache.v.copy_(ogc.value_cache.permute(0, 2, 1, 3))
    ids = torch.randint(3, 32000, (1, gamma + 1), generator=gen)
    pos = torch.arange(3000, 3000 + gamma + 1).unsqueeze(0)
    want = ot.forward(ids, M.FullCache(scfg, 8), ogc, position_ids=pos, spec=True)
    got = llm.retrieval_inference(ids, pos)
    assert torch.equal(got, want)
    S = 100
    okv = M.FullCache(scfg, S + 32)
    okv.key_cache[0, :S] = torch.randn(S, 2, 64, generator=gen).half()
    okv.value_cache[0, :S] = torch.randn(S, 2, 64, generator=gen).half()
    okv.seq_len = S
    llm.kv_cache.k[0, :, :S] = okv.key_cache[0, :S].permute(1, 0, 2)
    llm.kv_cache.v[0, :, :S] = okv.value_cache[0, :S].permute(1, 0, 2)
    llm.kv_cache.seq_len = S
    ids_t = torch.randint(3, 32000, (1, gamma + 2), generator=gen)
    assert torch.equal(llm.inference(ids_t), ot.forward(ids_t, okv, None))


def test_fp64_restatement_brackets_the_oracle():
    """The fp64-accumulating restatement used as 'truth' on the device (same fp16 rounding points as the oracle) stays
    within 2 fp16 spacings of the oracle, and the oracle is itself off the exactly-accumulated value at a sizeable share
    of the logits — the premise of the resolution-aware logit bar."""
    from tests.test_gpu_configs import _truth_retrieval_forward
    cfg = _cfg(hidden=1024, inter=2816, heads=8)
    cfg["max_position_embeddings"] = 131072
    cfg["rope_scaling"] = dict(type="yarn", factor=32.0, original_max_position_embeddings=4096)
    sd = specs.random_state_dict(cfg, 31)
    gamma, budget, prefill = 6, 512, 2048
    ot = M.OracleTarget(cfg, sd)
    ogc = M.RetrievalCacheO(cfg, budget, prefill, 8, gamma)
    gen = torch.Generator().manual_seed(5)
    ogc.key_cache.copy_(torch.randn(ogc.key_cache.shape, generator=gen).half())
    ogc.value_cache.copy_(torch.randn(ogc.value_cache.shape, generator=gen).half())
    gk0, gv0 = ogc.key_cache[0].clone(), ogc.value_cache[0].clone()
    ids = torch.randint(3, 32000, (1, gamma + 1), generator=gen)
    pos = torch.arange(100000, 100000 + gamma + 1).unsqueeze(0)
    oracle = ot.forward(ids, M.FullCache(cfg, 64), ogc, position_ids=pos, spec=True)
    truth = _truth_retrieval_forward(cfg, sd, ids, pos, gk0, gv0, ogc.real_budget - gamma - 1)
    d = (oracle - truth).abs()
    spacing = 2.0 ** (math.floor(math.log2(max(float(truth.abs().max()), 1.0))) - 10)
    assert float(d.max()) <= 2 * spacing and float(d.mean()) < 1e-3
    assert float((d > 0).float().mean()) > 0.05
    # the rows the oracle appended are the rows the restatement appended
    s = ogc.real_budget - gamma - 1
    assert float((ogc.key_cache[0, s:].float() - 0).abs().sum()) > 0
