"""Host logic of triforce_amd (caches, engine routing, decode loops) on CPU, with every HIP op swapped
for its oracle restatement (cpu_ops fixture).  Checks the product's control flow reproduces the
reference's golden token streams — nothing here measures or validates a kernel."""
import pytest
import torch

from oracle import ref_model as M
from tests import helpers as Hh


@pytest.mark.parametrize("name", ["small_gamma6", "cfg1_greedy"])
def test_triforce_greedy_matches_golden(cpu_ops, name):
    from triforce_amd.utils.decoding import Autoregressive, TriForce
    g = Hh.load_golden(name)
    ge = Hh.build_product(g, "cpu")
    prompt = Hh.prompt_of(g)
    tok = Hh.FakeTokenizer()
    _, ar = Autoregressive(tok, ge, prompt, max_len=g["gen_len"], top_k=-1, top_p=g["top_p"],
                           temperature=g["temperature"], return_tokens=True)
    assert ar == g["ar_tokens"]
    for rep in range(g["repeats"]):
        res = TriForce(tok, ge, prompt, gamma=g["gamma"], max_len=g["gen_len"], top_k=-1, top_p=g["top_p"],
                       temperature=g["temperature"], return_details=True)
        ref = g["triforce"][rep]
        assert res["tokens"] == ref["tokens"]
        assert abs(res["acceptance_rate"] - ref["acceptance_rate"]) < 1e-12
        assert res["counts"] == g["counts"][rep]
        assert ge.engine.kv_cache.seq_len == ref["final_seq_len"]
        assert ge.engine.draft_cache.seq_len == ref["draft_seq_len"]


def test_triforce_stochastic_matches_oracle_with_injected_uniforms(cpu_ops):
    """T=0.6 / top_p=0.9 (cfg3-style): product and oracle consume the same explicit uniform stream."""
    from triforce_amd.utils.decoding import TriForce
    from triforce_amd.utils.sampling import UniformSource
    g = Hh.load_golden("small_gamma6")
    us = Hh.fixed_uniforms()
    oeng, tsd, dsd = Hh.build_oracle(g, temperature=0.6, top_p=0.9)
    prompt = Hh.prompt_of(g)
    want = M.triforce(oeng, prompt, g["gamma"], 24, 0.6, 0.9, rng=M.InjectedRng(us))
    ge = Hh.build_product(g, "cpu", tsd, dsd, temperature=0.6, top_p=0.9)
    got = TriForce(Hh.FakeTokenizer(), ge, prompt, gamma=g["gamma"], max_len=24, top_k=-1, top_p=0.9, temperature=0.6,
                   rng=UniformSource("cpu", values=us), return_details=True)
    assert got["tokens"] == want["tokens"]
    assert got["counts"] == want["counts"]
    assert abs(got["acceptance_rate"] - want["acceptance_rate"]) < 1e-12
    assert want["accepted"] > 0      # the stochastic path actually accepts something


def test_product_logits_equal_oracle_logits(cpu_ops):
    """Same weights, fused qkv / gate-up GEMMs and fused residual+norm must not change a bit on CPU."""
    g = Hh.load_golden("forward_small")
    from oracle import specs
    from triforce_amd.models.cache import FlashSimpleCache
    from triforce_amd.models.config_yarn import LlamaConfig
    from triforce_amd.models.modeling_llama import LlamaForCausalLM
    tsd = specs.random_state_dict(g["tcfg"], g["tseed"])
    prompt = specs.random_prompt(512, 200, g["pseed"])
    m = LlamaForCausalLM.from_state_dict(LlamaConfig.from_dict(g["tcfg"]), tsd, "cpu")
    cache = FlashSimpleCache(m, 256)
    l1 = m(input_ids=prompt[:, :128], kv_cache=cache).logits
    l2 = m(input_ids=prompt[:, 128:200], kv_cache=cache).logits
    l3 = m(input_ids=prompt[:, :5], kv_cache=cache).logits
    assert torch.equal(l1[:, -1], g["target_logits"][0])
    assert torch.equal(l2[:, -1], g["target_logits"][1])
    assert torch.equal(l3, g["target_logits"][2])


def test_periodic_retrieval_rebuild_is_lossless_and_consistent(cpu_ops):
    """--rebuild_every: the retrieval cache only steers drafting, so greedy TriForce must still emit
    the target's greedy continuation, and after a rebuild every retrieved slot must hold the chunk its index names."""
    from triforce_amd.utils.decoding import TriForce
    g = Hh.load_golden("small_gamma6")
    ge = Hh.build_product(g, "cpu")
    prompt, tok = Hh.prompt_of(g), Hh.FakeTokenizer()
    res = TriForce(tok, ge, prompt, gamma=g["gamma"], max_len=g["gen_len"], top_k=-1, top_p=g["top_p"],
                   temperature=g["temperature"], return_details=True, rebuild_every=2)
    n = min(len(res["tokens"]), len(g["ar_tokens"]))
    assert res["tokens"][:n] == g["ar_tokens"][:n]                       # lossless
    rc, kv = ge.engine.graph_cache, ge.engine.kv_cache
    assert res["outer_steps"] >= 2
    gen = kv.seq_len - rc.prefill                                        # generated tail occupies [B - gen, B)
    cs = rc.chunk_size
    for layer in (0, rc.layers - 1):
        idx = rc.last_idx[layer]                                         # (H, sets), chunk 0 first
        assert (idx[:, 0] == 0).all()
        src_k, _ = kv.layer_kv(layer)
        for h in range(rc.num_heads):
            for j in range(min(rc.select_sets, (rc.max_budget - gen) // cs)):
                c = int(idx[h, j])
                assert torch.equal(rc.k[layer, h, j * cs:(j + 1) * cs], src_k[h, c * cs:(c + 1) * cs]), (layer, h, j)
    # the selection did move with the query (otherwise this test would not exercise the rebuild)
    ge2 = Hh.build_product(g, "cpu")
    TriForce(tok, ge2, prompt, gamma=g["gamma"], max_len=g["gen_len"], top_k=-1, top_p=g["top_p"],
             temperature=g["temperature"], return_details=True)
    assert any(not torch.equal(rc.last_idx[i], ge2.engine.graph_cache.last_idx[i]) for i in range(rc.layers))


def _tp_product(g, tsd, dsd, temperature, top_p):
    from triforce_amd.models.cache import StreamingLLMEvictionCache
    from triforce_amd.models.config_yarn import LlamaConfig
    from triforce_amd.models.modeling_llama_68m import LlamaForCausalLM as Draft
    from triforce_amd.models.TP_llama import DistributedLlama
    gamma = g["gamma"]
    draft = Draft.from_state_dict(LlamaConfig.from_dict(g["dcfg"]), dsd, "cpu")
    dcache = StreamingLLMEvictionCache(draft, start_size=16, recent_size=256 - 16 - gamma, gamma=gamma)
    tcfg = LlamaConfig.from_dict(g["tcfg"])
    llm = DistributedLlama("unused", config=tcfg, device="cpu", local_rank=0, world_size=1, prefill=g["prefill"],
                           gen_len=g["gen_len"], temperature=temperature, top_p=top_p, retrieval_budget=g["budget"],
                           retrieval_chunk_size=g["chunk"], kv_offload=True, on_chip_layers=tcfg.num_hidden_layers,
                           draft=draft, draft_cache=dcache, gamma=gamma)
    llm.init_parameters(tsd)
    return llm


@pytest.mark.parametrize("name", ["tp_chain", "tp_chain_gamma16"])
@pytest.mark.parametrize("temperature,top_p", [(0.6, 0.9), (1.0, 1e-9)])
def test_tp_chain_loop_matches_oracle_with_injected_uniforms(cpu_ops, temperature, top_p, name):
    """TriForce_Dist on the product's TP engine (world size 1) against the restatement that
    tests/golden/tp_chain.pt pins to the reference's TriForce_Dist: same explicit uniforms -> same stream, accept
    counts, returned average and final cache lengths.  Covers the TP-only rules: inclusive outer accept, draft
    sampled at 0.6 / 0.9 whatever the target's settings, 128-token draft prefill blocks."""
    from triforce_amd.utils.decoding import TriForce_Dist
    from triforce_amd.utils.sampling import UniformSource
    g = Hh.load_golden(name)
    us = Hh.fixed_uniforms(seed=21)
    prompt = Hh.prompt_of(g)
    oeng, tsd, dsd = Hh.build_oracle_tp(g, temperature, top_p)
    want = M.triforce(oeng, prompt, g["gamma"], 30, temperature, top_p, rng=M.InjectedRng(us), eos_token_id=-1, dist=True)
    llm = _tp_product(g, tsd, dsd, temperature, top_p)
    tok = Hh.FakeTokenizer()
    tok.eos_token_id = -1
    got = TriForce_Dist(tok, llm, prompt, gamma=g["gamma"], max_len=30, top_k=-1, top_p=top_p, temperature=temperature,
                        rng=UniformSource("cpu", values=us), return_details=True)
    assert got["tokens"] == want["tokens"] and got["counts"] == want["counts"]
    assert abs(got["avg_tokens"] - want["avg_tokens"]) < 1e-12
    assert llm.kv_cache.seq_len == oeng.kv_cache.seq_len and llm.draft_cache.seq_len == oeng.draft_cache.seq_len


def test_tp_chain_eos_exits_like_the_reference(cpu_ops):
    """decoding.py:357-360 / :382-383: the TP loop ends when the token that CLOSED the accept scan (an accepted or a
    resampled token) is eos — not when a bonus token is (that one is sampled after the check).  Every token of the
    first steps is tried as eos, so accepted, resampled and bonus placements all occur."""
    from triforce_amd.utils.decoding import TriForce_Dist
    from triforce_amd.utils.sampling import UniformSource
    g = Hh.load_golden("tp_chain")
    us = Hh.fixed_uniforms(seed=21)
    prompt = Hh.prompt_of(g)
    oeng, tsd, dsd = Hh.build_oracle_tp(g, 0.6, 0.9)
    base = M.triforce(oeng, prompt, g["gamma"], 24, 0.6, 0.9, rng=M.InjectedRng(us), eos_token_id=-1, dist=True)
    assert any(c >= 2 for c in base["counts"]), "no accepted draft in the probe stream"
    kinds = set()
    for pick in sorted(set(base["tokens"][1:16])):
        oeng, _, _ = Hh.build_oracle_tp(g, 0.6, 0.9)                 # fresh engines: a second prompt carries the
        llm = _tp_product(g, tsd, dsd, 0.6, 0.9)                     # draft cache's seq_len over (reference quirk)
        want = M.triforce(oeng, prompt, g["gamma"], 24, 0.6, 0.9, rng=M.InjectedRng(us), eos_token_id=pick, dist=True)
        tok = Hh.FakeTokenizer()
        tok.eos_token_id = pick
        got = TriForce_Dist(tok, llm, prompt, gamma=g["gamma"], max_len=24, top_k=-1, top_p=0.9, temperature=0.6,
                            rng=UniformSource("cpu", values=us), return_details=True)
        assert got["tokens"] == want["tokens"], (pick, got["tokens"], want["tokens"])
        assert got["accepted"] == want["accepted"] and got["drafted"] == want["drafted"], pick
        kinds.add("stopped" if want["tokens"][-1] == pick and len(want["tokens"]) < len(base["tokens"]) else "ran on")
    assert kinds == {"stopped", "ran on"}, kinds          # both behaviours were exercised


def test_tp_baseline_matches_oracle_with_injected_uniforms(cpu_ops):
    """Baseline_Dist (decoding.py:243-287; pinned to the reference in tests/golden/tp_chain.pt): whole prompt through
    the 128-token prefill, then one token per forward; returns the tokens AFTER the first sampled one."""
    from triforce_amd.utils.decoding import Baseline_Dist
    from triforce_amd.utils.sampling import UniformSource
    g = Hh.load_golden("tp_chain")
    us = Hh.fixed_uniforms(seed=23)
    prompt = Hh.prompt_of(g)
    oeng, tsd, dsd = Hh.build_oracle_tp(g, 0.6, 0.9)
    want = M.autoregressive(oeng, prompt, 16, 0.6, 0.9, rng=M.InjectedRng(us))
    llm = _tp_product(g, tsd, dsd, 0.6, 0.9)
    ms, got = Baseline_Dist(Hh.FakeTokenizer(), llm, prompt, max_len=16, top_k=-1, top_p=0.9, temperature=0.6,
                            rng=UniformSource("cpu", values=us))
    assert got[0].tolist() == want[1:] and ms > 0
    assert llm.kv_cache.seq_len == oeng.kv_cache.seq_len == g["prefill"] + 16
