"""The CPU oracle (oracle/ref_ops.py, oracle/ref_model.py) against the golden fixtures that
oracle/gen_golden.py recorded from the UNMODIFIED reference (tests/golden/*.pt).  This is what pins
the oracle: the reference itself ships no golden vectors for this path."""
import os

import pytest
import torch

from oracle import ref_model as M
from oracle import ref_ops as R
from oracle import specs
from tests import helpers as Hh


def test_rope_tables_match_reference():
    g = Hh.load_golden("rope_tables")
    cos, sin = R.rope_tables_yarn(128, 131072, 32.0, 4096)
    assert torch.equal(cos[g["rows"]], g["yarn_cos"]) and torch.equal(sin[g["rows"]], g["yarn_sin"])
    pc, ps = R.rope_tables_plain(128, 131072, 1e7)
    assert torch.equal(pc[g["rows"]], g["plain_cos"]) and torch.equal(ps[g["rows"]], g["plain_sin"])
    assert abs(g["yarn_mscale"] - 1.34657) < 1e-5


def test_forward_and_sampling_match_reference():
    g = Hh.load_golden("forward_small")
    tsd, dsd = specs.random_state_dict(g["tcfg"], g["tseed"]), specs.random_state_dict(g["dcfg"], g["dseed"])
    prompt = specs.random_prompt(512, 200, g["pseed"])
    ot, okv = M.OracleTarget(g["tcfg"], tsd), M.FullCache(g["tcfg"], 256)
    assert torch.equal(ot.forward(prompt[:, :128], okv)[:, -1], g["target_logits"][0])
    assert torch.equal(ot.forward(prompt[:, 128:200], okv)[:, -1], g["target_logits"][1])
    assert torch.equal(ot.forward(prompt[:, :5], okv), g["target_logits"][2])
    od, odc = M.OracleDraft(g["dcfg"], dsd), M.StreamingCacheO(g["dcfg"], gamma=4, start_size=16, recent_size=100)
    for i in range(4):
        odc.evict_prefill(50)
        last = od.forward(prompt[:, i * 50:(i + 1) * 50], odc, None)
    assert torch.equal(last[:, -1], g["draft_prefill_last"])
    assert torch.equal(od.forward(prompt[:, :3], odc, odc, gamma_offset=2), g["draft_spec_logits"])
    for (T, P), want in g["sampling_probs"].items():
        assert torch.equal(R.norm_logits(g["sampling_logits"].clone(), T, -1, P), want)
    assert torch.equal(R.max_fn(g["maxfn_in"]), g["maxfn_out"])


@pytest.mark.parametrize("name", ["small_gamma6", "cfg1_greedy", "cfg1_stochastic"])
def test_triforce_streams_match_reference(name):
    g = Hh.load_golden(name)
    eng, _, _ = Hh.build_oracle(g)
    prompt = Hh.prompt_of(g)
    if g["rng_seed"] is not None:
        torch.manual_seed(g["rng_seed"])
    assert M.autoregressive(eng, prompt, g["gen_len"], g["temperature"], g["top_p"]) == g["ar_tokens"]
    for rep in range(g["repeats"]):
        if g["rng_seed"] is not None:
            torch.manual_seed(g["rng_seed"])
        res = M.triforce(eng, prompt, g["gamma"], g["gen_len"], g["temperature"], g["top_p"])
        assert res["tokens"] == g["triforce"][rep]["tokens"]
        assert abs(res["acceptance_rate"] - g["triforce"][rep]["acceptance_rate"]) < 1e-12
    # stage-wise: retrieval scores bit-identical to the reference, top-k equal up to the tie order
    scores = torch.stack(eng.graph_cache.last_scores)
    assert torch.equal(scores[:, :, 1:], g["retrieval_scores"])
    for l in range(scores.shape[0]):
        assert R.topk_matches_reference(scores[l], eng.graph_cache.last_idx[l], g["retrieval_idx"][l])
    if g["temperature"] == 1.0:
        n = min(len(g["ar_tokens"]), len(g["triforce"][0]["tokens"]))
        assert g["triforce"][0]["tokens"][:n] == g["ar_tokens"][:n]     # lossless: greedy TriForce == greedy AR


@pytest.mark.parametrize("name", ["tp_chain", "tp_chain_gamma16"])
def test_tp_chain_matches_reference(name):
    """TriForce_Dist / Middle_Spec_Dist (decoding.py:291-495) on the reference's TP engine (TP_llama.py), run
    unmodified on CPU when the golden was made: the restatement reproduces streams, accept counts and the returned
    average for a stochastic target, a greedy target (stochastic draft) and two eos placements."""
    g = Hh.load_golden(name)                      # gamma 6, and gamma 16 (the README command of offloading_TP.py)
    prompt = Hh.prompt_of(g)
    for case in g["cases"]:
        eng, _, _ = Hh.build_oracle_tp(g, case["temperature"], case["top_p"])
        torch.manual_seed(case["rng_seed"])
        res = M.triforce(eng, prompt, g["gamma"], g["gen_len"], case["temperature"], case["top_p"],
                         eos_token_id=case["eos"], dist=True)
        assert res["tokens"] == case["tokens"], case["label"]
        assert res["counts"][:len(case["counts"])] == case["counts"], case["label"]
        assert abs(res["avg_tokens"] - case["avg_tokens"]) < 1e-12, case["label"]
        if not case["ended_by_eos"]:
            assert eng.kv_cache.seq_len == case["final_seq_len"] and eng.draft_cache.seq_len == case["draft_seq_len"]
        else:
            assert res["tokens"][-1] == case["eos"] and len(res["tokens"]) < g["gen_len"]
    for b in g["baselines"]:                                   # Baseline_Dist (decoding.py:243-287)
        eng, _, _ = Hh.build_oracle_tp(g, b["temperature"], b["top_p"])
        torch.manual_seed(b["rng_seed"])
        out = M.autoregressive(eng, prompt, len(b["gen_tokens"]), b["temperature"], b["top_p"])
        assert out[1:] == b["gen_tokens"], b["label"]


def test_offloading_entry_stream_matches_reference():
    """test/offloading.py's configuration (OffloadingFlashSimpleCache, capacity prefill + gen_len + 32) run with the
    unmodified reference on CPU: recorded stream == the restatement's, and (asserted when the fixture was made) ==
    the FlashSimpleCache stream of the same seed — offloading moves the KV, it does not change a token."""
    import json
    import os
    g = json.load(open(os.path.join(Hh.GOLDEN, "offloading_small.json")))
    assert g["equals_resident_stream"] is True
    tsd = Hh.specs.random_state_dict(g["tcfg"], g["tseed"], head_std=g["head_std"])
    dsd = Hh.specs.random_state_dict(g["dcfg"], g["dseed"], head_std=g["head_std"])
    gamma = g["gamma"]
    eng = M.OracleEngine(M.OracleTarget(g["tcfg"], tsd), M.FullCache(g["tcfg"], g["prefill"] + g["gen_len"] + 32),
                         M.RetrievalCacheO(g["tcfg"], g["budget"], g["prefill"], g["chunk"], gamma),
                         M.OracleDraft(g["dcfg"], dsd),
                         M.StreamingCacheO(g["dcfg"], gamma=gamma, start_size=16, recent_size=256 - 16 - gamma),
                         g["temperature"], g["top_p"])
    torch.manual_seed(g["rng_seed"])
    res = M.triforce(eng, Hh.prompt_of(g), gamma, g["gen_len"], g["temperature"], g["top_p"], eos_token_id=-1)
    assert res["tokens"] == g["tokens"] and abs(res["acceptance_rate"] - g["acceptance_rate"]) < 1e-12
    assert eng.kv_cache.seq_len == g["final_seq_len"]


def test_topk_canonical_tie_rule():
    s = torch.tensor([[9.0, 1, 3, 3, 2, 3, -1, 3]], dtype=torch.float16)
    assert R.retrieval_topk(s, 4).tolist() == [[0, 2, 3, 5]]           # ties -> lowest chunk first
    assert R.retrieval_topk(s, 8).tolist() == [[0, 2, 3, 5, 7, 4, 1, 6]]


def test_accept_chain_edge_cases():
    V = 16
    p = torch.full((4, V), 1.0 / V)
    q = torch.full((3, V), 1.0 / V)
    q[1, 5] = 0.0                                                       # p/q = inf -> min(1,inf)=1 -> accept
    p[2, 7] = 0.0
    q[2, 7] = 0.0                                                       # 0/0 = nan -> `r < nan` is False -> reject
    count, flags = R.accept_chain(p, q, [1, 5, 7], [0.5, 0.999, 0.0])
    assert flags == [True, True, False] and count == 2
    assert R.accept_chain(p, q, [1, 2, 3], [1.0, 0.0, 0.0], inclusive=True)[0] == 3    # r<=1 (TP variant)
    assert R.accept_chain(p, q, [1, 2, 3], [1.0, 0.0, 0.0], inclusive=False)[0] == 0
    c, nxt, reason, used = R.accept_and_correct(p, q, [1, 2, 3], [0.1, 0.1, 0.1, 0.5], eos_token_id=2)
    assert (c, nxt, reason, used) == (2, 2, 2, 2)                       # accepted eos stops the chain, no resample
    c, nxt, reason, used = R.accept_and_correct(p, q, [1, 3, 2], [0.1, 0.1, 0.1, 0.5], eos_token_id=2)
    assert (c, reason, used) == (3, 1, 4)                               # eos as the LAST token: bonus path still runs


def test_shard_of_matches_the_reference_slicing_for_every_rank():
    """tests/golden/tp_shards.json: digests of the shards the UNMODIFIED reference cut (TP_layers.py:126-147,
    DistributedLlamaLayer.init_parameters) for every rank of a 4-way and an 8-way split of one seeded layer with the 13B's
    proportions (40 heads, intermediate 1728 = 8 x 216).  oracle.specs.shard_of — the oracle side of the 13B / TP = 8
    shard-width GPU parity test — must cut the very same tensors."""
    import json
    g = json.load(open(os.path.join(Hh.GOLDEN, "tp_shards.json")))
    cfg = g["cfg"]
    sd = specs.random_state_dict(cfg, g["seed"])
    pre = "model.layers.0."
    names = dict(q="self_attn.q_proj", k="self_attn.k_proj", v="self_attn.v_proj", o="self_attn.o_proj", gate="mlp.gate_proj",
                 up="mlp.up_proj", down="mlp.down_proj")
    assert len(g["cases"]) == 12
    for case in g["cases"]:
        scfg, ssd = specs.shard_of(cfg, sd, case["rank"], case["world"])
        assert scfg["head_dim"] == cfg["hidden_size"] // cfg["num_attention_heads"]
        for short, full in names.items():
            t = ssd[pre + full + ".weight"]
            want = case["shards"][short]
            assert list(t.shape) == want["shape"], (case["world"], case["rank"], short)
            assert specs.tensor_digest(t) == want["sha256"], (case["world"], case["rank"], short)
