"""Aligned synthetic weights (triforce_amd/models/aligned.py) on CPU: the planted table, the exactness of the planted
logits, the calibrated retrieval -> target acceptance, and the acceptance the decode loop then actually sees (host logic
of the product over the oracle-backed CPU ops; the oracle is the checker here, never the product path)."""
import math

import pytest
import torch

from oracle import specs
from triforce_amd.models import aligned
from triforce_amd.models.config_yarn import LlamaConfig

V = 4096


def _cfgs():
    tcfg = LlamaConfig.from_dict(specs.tiny_target_config(vocab_size=V, layers=2, hidden=256, heads=2, max_pos=8192))
    dcfg = LlamaConfig.from_dict(specs.draft_68m_config(vocab_size=V))
    return tcfg, dcfg


def test_spec_parsing():
    assert aligned.parse_spec("random:1") is None and aligned.parse_spec("/some/dir") is None
    s = aligned.parse_spec("aligned")
    assert (s.draft_acc, s.retrieval_acc, s.seed) == (0.7, 0.9, 0)
    s = aligned.parse_spec("aligned:0.5:0.95:7")
    assert (s.draft_acc, s.retrieval_acc, s.seed) == (0.5, 0.95, 7) and s.label() == "aligned:0.5:0.95:7"
    with pytest.raises(ValueError):
        aligned.parse_spec("aligned:1.5")


def test_planted_tables():
    spec = aligned.parse_spec("aligned:0.6:0.8:3")
    tab = aligned.plant(V, spec)
    st, sd, off, agree = tab["succ_t"], tab["succ_d"], tab["off"], tab["agree"]
    K = spec.candidates
    assert st.shape == (K, V) and int(st.min()) >= 3 and int(st.max()) < V          # never a special token (eos = 2)
    for a in range(K):
        for b in range(a + 1, K):
            assert not bool((st[a] == st[b]).any())                                   # distinct successors per token
        assert st[a, 3:].unique().numel() == V - 3                                    # each token is a k-th successor once
    assert bool((off[0] == 0).all()) and bool((off[1:] < 0).all()) and bool((off[2] < off[1]).all())
    assert abs(float(agree.float().mean()) - 0.6 / 0.8) < 0.03                        # planted fraction = draft / retrieval
    assert bool((sd[:, agree] == st[:, agree]).all())
    assert float((sd[0, ~agree] == st[0, ~agree]).float().mean()) < 0.01               # chance coincidences only
    again = aligned.plant(V, spec)
    assert bool((again["succ_t"] == st).all()) and bool((again["off"] == off).all())  # a function of (vocab, spec) only


@pytest.mark.parametrize("role", ["target", "draft"])
def test_planted_logits_are_exact(role):
    """fp16(lm_head) . rmsnorm(embed[t]) hits peak + off at every planted pair, everything else stays far below."""
    from triforce_amd.models.llama_core import LlamaWeights
    tcfg, dcfg = _cfgs()
    cfg = tcfg if role == "target" else dcfg
    spec = aligned.parse_spec("aligned:0.7:0.9:1")
    W = LlamaWeights(cfg, "cpu").init_aligned(spec, role, attn_keys=256)
    tab = aligned.plant(V, spec)
    succ = tab["succ_t"] if role == "target" else tab["succ_d"]
    E, head = W.embed, W.lm_head.w
    n = aligned._rms_rows(E, W.eps).float()
    L = (n @ head.float().T).to(torch.float16).float()
    rows = torch.arange(V)
    for k in range(spec.candidates):
        got = L[rows, succ[k]]
        assert float((got - (spec.peak + tab["off"][k])).abs().max()) < 0.15, (role, k)
    mask = torch.ones_like(L, dtype=torch.bool)
    for k in range(spec.candidates):
        mask[rows, succ[k]] = False
    # background (cross-talk ~ peak * sqrt(2K / dim(S1)) per entry: S1 is only 128-wide for the tiny target, 2048 for
    # the 7B) never reaches the planted candidates, and holds a negligible share of the probability mass at T = 0.6
    assert float(L[mask].max()) < spec.peak - 2.0
    bg = torch.where(mask, L, torch.full_like(L, -1e9))
    share = torch.logsumexp(bg / 0.6, dim=1) - torch.logsumexp(L / 0.6, dim=1)
    assert float(share.exp().max()) < 0.05
    # dense, not zero-padded: every block of every matrix carries values
    s1, s2, s3 = aligned.subspaces(cfg.hidden_size)
    for m in (W.wo[0].w, W.wd[0].w):
        for sl in (s1, s2, s3):
            assert float((m[sl] == 0).float().mean()) < 0.2
    assert W.aligned["planted_fit_err"] < 0.15


def _engine(spec_str, P, B, gamma):
    from triforce_amd.models.cache import FlashSimpleCache, RetrievalCache, StreamingLLMEvictionCache
    from triforce_amd.models.modeling_llama import LlamaForCausalLM
    from triforce_amd.models.modeling_llama_68m import LlamaForCausalLM as Draft
    from triforce_amd.utils.graph_infer import GraphInferenceEngine
    tcfg, dcfg = _cfgs()
    spec = aligned.parse_spec(spec_str)
    target = LlamaForCausalLM(tcfg, "cpu").init_aligned(spec, attn_keys=B)
    draft = Draft(dcfg, "cpu").init_aligned(spec, attn_keys=256)
    ge = GraphInferenceEngine(target, FlashSimpleCache(target, P + 400),
                              RetrievalCache(target, max_budget=B, prefill=P, gamma=gamma, chunk_size=8), draft,
                              StreamingLLMEvictionCache(draft, start_size=16, recent_size=256 - 16 - gamma, gamma=gamma))
    ge.initialize_eager(gamma, probs=True, temperature=0.6, top_p=0.9)
    return ge, target


class _Tok:
    eos_token_id = 2


@pytest.mark.parametrize("spec_str,lo_mid,hi_mid,lo_tgt,hi_tgt", [
    ("aligned:0.7:0.9:0", 0.50, 0.85, 0.78, 0.97),
    ("aligned:0.3:0.97:0", 0.15, 0.45, 0.90, 1.00),
])
def test_decode_loop_sees_the_requested_acceptance(cpu_ops, spec_str, lo_mid, hi_mid, lo_tgt, hi_tgt):
    from triforce_amd.utils.decoding import TriForceRunner
    from triforce_amd.utils.sampling import UniformSource
    P, B, gamma = 2048, 256, 4
    ge, target = _engine(spec_str, P, B, gamma)
    run = TriForceRunner(_Tok(), ge, gamma, top_k=-1, top_p=0.9, temperature=0.6, rng=UniformSource("cpu", seed=3))
    run.prefill(specs.random_prompt(V, P, 11))                       # calibrates the read-out on the way
    cal = target.weights.aligned["calibration"]
    want = aligned.parse_spec(spec_str).retrieval_acc
    assert abs(cal["probe_acceptance"] - want) < 0.02 and cal["probe_acceptance_gain0"] > 0.98
    assert cal["readout_gain"] > 0 and cal["s2_difference_over_norm"] > 0
    steps = 40
    for _ in range(steps):
        run.step()
    tests = run.accepted_count + run.resample_count                  # accept tests of the target
    per_token = run.accepted_count / tests
    middle = sum(run.acc_rate_middle_list) / len(run.acc_rate_middle_list)
    assert lo_tgt <= per_token <= hi_tgt, per_token
    assert lo_mid <= middle <= hi_mid, middle
    assert run.n / steps > (2.0 if want < 0.95 else 2.5)             # several tokens per outer step, not ~1
    assert 2 not in run.emitted and all(3 <= t < V for t in run.emitted)


def test_overwrite_random_is_in_place_and_equals_init_random():
    from triforce_amd.models.llama_core import LlamaWeights
    tcfg, _ = _cfgs()
    a = LlamaWeights(tcfg, "cpu").init_aligned(aligned.parse_spec("aligned"), "target", attn_keys=256)
    ptrs = [a.lm_head.w.data_ptr(), a.wqkv[0].w.data_ptr(), a.wd[1].w.data_ptr(), a.embed.data_ptr()]
    a.overwrite_random_(5)
    b = LlamaWeights(tcfg, "cpu").init_random(5)
    assert ptrs == [a.lm_head.w.data_ptr(), a.wqkv[0].w.data_ptr(), a.wd[1].w.data_ptr(), a.embed.data_ptr()]
    assert a.aligned is None
    for x, y in ((a.lm_head.w, b.lm_head.w), (a.embed, b.embed), (a.wqkv[1].w, b.wqkv[1].w), (a.wo[0].w, b.wo[0].w),
                 (a.wgu[0].w, b.wgu[0].w), (a.wd[1].w, b.wd[1].w), (a.ln1[0], b.ln1[0]), (a.norm, b.norm)):
        assert torch.equal(x, y)
    assert math.isclose(float(a.wd[0].w.float().std()), 0.02, rel_tol=0.05)
