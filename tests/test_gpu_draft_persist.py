"""The one-launch draft forward (tf_draft_forward_68m_persist, csrc/draft_persist.hip) against the 13-launch chain
(tf_draft_forward_68m) on the real Llama-68M shape: logits, top-p probability rows and the K / V rows written must be
BIT-IDENTICAL — the chain itself is checked against the oracle (tests/test_gpu_e2e.py, test_gpu_ops.py), so this pins the
persistent form to the oracle at the same tolerance with nothing added.  Also: hipGraph replay, a soak of replays (a
stale hand-off would show as a mismatch), degenerate probability rows (ties at the top-p boundary, flat rows, top_p = 1),
and the failure path: a lost arrival must time out, poison the outputs, stay sticky and recover after a reset.
Reference: models/modeling_llama_68m.py:129-190, utils/graph_infer.py:52-57, utils/sampling.py:5-27,43-60."""
import ctypes

import pytest
import torch

from oracle import specs
from triforce_amd import hip, ops
from triforce_amd.models import zoo

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GAMMA = 6


def _cfg(vocab=32000):
    cfg = dict(zoo.CONFIGS["llama-68M"])
    cfg.update(vocab_size=vocab, num_key_value_heads=cfg["num_attention_heads"], rope_theta=10000.0, rope_scaling=None,
               hidden_act="silu")
    return cfg


def _build(sd, cfg, persist, monkeypatch):
    from triforce_amd.models.cache import StreamingLLMEvictionCache
    from triforce_amd.models.config_yarn import LlamaConfig
    from triforce_amd.models.modeling_llama_68m import LlamaForCausalLM as Draft
    monkeypatch.setattr(ops, "DRAFT_PERSIST", persist)
    m = Draft.from_state_dict(LlamaConfig.from_dict(cfg), sd, DEV)
    c = StreamingLLMEvictionCache(m, start_size=16, recent_size=256 - 16 - GAMMA, gamma=GAMMA)
    assert m._native_model() is not None
    assert (m._persist is not None) == persist
    return m, c


def _run(m, c, seed, settings=((0.6, 0.9),), vocab=32000):
    gen = torch.Generator().manual_seed(seed)
    outs = []
    for n in (16, 7, 1, 12):                                           # non-speculative appends (the cache fills up)
        ids = torch.randint(3, vocab, (1, n), generator=gen).to(DEV)
        outs.append(m.forward(ids, c, None, -1).logits)
    for T, top_p in settings:
        for off in range(GAMMA + 3):                                   # speculative steps at every offset, with top-p
            ids = torch.randint(3, vocab, (1, off + 1), generator=gen).to(DEV)
            o = m.forward(ids, c, c, off, probs=(T, top_p))
            outs += [o.logits, o.probs]
    torch.cuda.synchronize()
    return outs


def _same(ref, got):
    assert len(ref) == len(got)
    for i, (a, b) in enumerate(zip(ref, got)):
        assert a.shape == b.shape
        if not torch.equal(a, b):
            d = (a.float() - b.float()).abs()
            raise AssertionError(f"output {i}: {int((a != b).sum())} of {a.numel()} entries differ, max |d| {float(d.max()):.3e}")


@pytest.fixture(scope="module")
def sd68():
    return specs.random_state_dict(_cfg(), 11, head_std=0.05)


def test_one_launch_draft_is_bit_identical_to_the_chain(sd68, monkeypatch):
    cfg = _cfg()
    settings = ((0.6, 0.9), (1.0, 0.95), (0.6, 1.0), (0.3, 0.5))
    m0, c0 = _build(sd68, cfg, False, monkeypatch)
    ref = _run(m0, c0, 5, settings)
    m1, c1 = _build(sd68, cfg, True, monkeypatch)
    got = _run(m1, c1, 5, settings)
    _same(ref, got)
    assert torch.equal(c0.k, c1.k) and torch.equal(c0.v, c1.v) and c0.seq_len == c1.seq_len
    assert m1._persist.error() == 0
    # the probability rows are real distributions
    for p in got[5::2]:
        assert abs(float(p.sum()) - 1.0) < 1e-4 and float(p.min()) >= 0.0


def test_one_launch_draft_graph_replay_and_soak(sd68, monkeypatch):
    """Captured once per speculative offset, replayed 300 times each in rotation over changing tokens: every replay equals
    the chain's result for the same tokens (the launch epoch advances under replay; nothing is re-zeroed in between)."""
    cfg = _cfg()
    m0, c0 = _build(sd68, cfg, False, monkeypatch)
    m1, c1 = _build(sd68, cfg, True, monkeypatch)
    _same(_run(m0, c0, 7), _run(m1, c1, 7))
    gen = torch.Generator().manual_seed(3)
    offs = (0, 2, 5, GAMMA + 2)
    toks = [torch.randint(3, 32000, (4, off + 1), generator=gen).to(DEV) for off in offs]
    want = []
    for off, tk in zip(offs, toks):
        monkeypatch.setattr(ops, "DRAFT_PERSIST", False)
        want.append([m0.forward(tk[i:i + 1], c0, c0, off, probs=(0.6, 0.9)) for i in range(4)])
    monkeypatch.setattr(ops, "DRAFT_PERSIST", True)
    graphs = []
    for off, tk in zip(offs, toks):
        buf = tk[0:1].clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            m1.forward(buf, c1, c1, off, probs=(0.6, 0.9))
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            cap = m1.forward(buf, c1, c1, off, probs=(0.6, 0.9))
        graphs.append((g, buf, cap))
    for it in range(300):
        for k, (g, buf, cap) in enumerate(graphs):
            i = (it + k) % 4
            buf.copy_(toks[k][i:i + 1])
            g.replay()
            if it % 25 == 0 or it == 299:
                torch.cuda.synchronize()
                assert torch.equal(cap.logits, want[k][i].logits) and torch.equal(cap.probs, want[k][i].probs), (it, k)
    torch.cuda.synchronize()
    assert m1._persist.error() == 0


@pytest.mark.parametrize("kind", ["ties", "ties8", "ties20", "mid", "flat", "peaked"])
def test_one_launch_top_p_on_degenerate_rows(kind, monkeypatch):
    """Rows the exact select must get right bit for bit: `ties` — lm_head with 16 distinct rows repeated (2 000 entries share
    every logit: the boundary falls inside a tie group and the ties are ranked by index — by the radix select on the index,
    streamed when the candidates overflow the list; `ties8`: tie groups of 8; `ties20`: 20 equal dominant logits, top_p cuts
    the group — the in-wave finish ranks them), `mid` — logits of a few units' spread (hundreds of candidates, a boundary bin
    of more or fewer than 64 entries depending on the setting), `flat` — lm_head = 0 (32 000 equal entries), `peaked` — one
    dominant logit (a trained draft's usual row)."""
    cfg = _cfg()
    sd = dict(specs.random_state_dict(cfg, 23, head_std=0.05))
    head = sd["lm_head.weight"].clone()
    if kind == "ties":
        head = head[:16].repeat(2000, 1)
    elif kind == "ties8":                                              # 4 000 tie groups of 8, sharper logits: the boundary cuts a
        head = (head[:4000] * 4.0).repeat_interleave(8, dim=0)         # group that fits the boundary-bin list (ranked through it)
    elif kind == "ties20":
        head[100:120] = head[777] * 40.0
    elif kind == "mid":
        head *= 12.0
    elif kind == "flat":
        head.zero_()
    else:
        head[777] *= 40.0
    sd["lm_head.weight"] = head
    settings = ((0.6, 0.9), (1.0, 0.5), (0.8, 1.0))
    m0, c0 = _build(sd, cfg, False, monkeypatch)
    ref = _run(m0, c0, 9, settings)
    m1, c1 = _build(sd, cfg, True, monkeypatch)
    got = _run(m1, c1, 9, settings)
    _same(ref, got)
    assert m1._persist.error() == 0


def test_one_launch_draft_small_vocab_and_one_layer(monkeypatch):
    """Shapes at the edges of what the launch takes: a vocabulary that leaves the last workgroups without a panel (V = 8 208:
    513 panels on 2 048 wave slots — the smallest the chain runs with the 4-way K split this form reproduces), the largest
    (32 768), and a one-layer model."""
    for vocab, layers in ((8208, 2), (32000, 1), (32768, 2)):
        cfg = _cfg(vocab)
        cfg["num_hidden_layers"] = layers
        sd = specs.random_state_dict(cfg, 31 + layers, head_std=0.05)
        m0, c0 = _build(sd, cfg, False, monkeypatch)
        ref = _run(m0, c0, 13, vocab=vocab)
        m1, c1 = _build(sd, cfg, True, monkeypatch)
        got = _run(m1, c1, 13, vocab=vocab)
        _same(ref, got)
        assert torch.equal(c0.k, c1.k) and torch.equal(c0.v, c1.v)


def test_one_launch_draft_unsupported_shapes_keep_the_chain(sd68, monkeypatch):
    m1, c1 = _build(sd68, _cfg(), True, monkeypatch)
    L = hip.lib()
    nm = m1._native_model()
    assert L.tf_draft_persist_supported(ctypes.byref(nm), 16, 300) == 0
    assert L.tf_draft_persist_supported(ctypes.byref(nm), 17, 300) == -22          # two row tiles: the chain
    assert L.tf_draft_persist_supported(ctypes.byref(nm), 4, 385) == -22
    small = hip.TfDraftModel.from_buffer_copy(nm)
    small.vocab = 8192                                                             # 512 panels: the chain splits K 8 ways there
    assert L.tf_draft_persist_supported(ctypes.byref(small), 4, 100) == -22
    assert L.tf_draft_persist_supported(None, 4, 100) == -22
    ids = torch.randint(3, 32000, (1, 20)).to(DEV)                                 # 20 rows: runs (through the chain)
    out = m1.forward(ids, c1, None, -1).logits
    assert out.shape == (1, 20, 32000) and bool(torch.isfinite(out).all())


def test_one_launch_draft_lost_arrival_times_out_poisons_and_recovers(sd68, monkeypatch):
    """Fault injection (tf_draft_persist_tune key 1): the first producer of an edge loses its arrival.  The waiters must give
    up after the wall-clock limit instead of hanging the GPU, the error word names the edge, logits and probabilities are
    NaN, the block stays failed for later launches, and after tf_draft_persist_reset results are bit-identical again."""
    L = hip.lib()
    cfg = _cfg()
    m0, c0 = _build(sd68, cfg, False, monkeypatch)
    m1, c1 = _build(sd68, cfg, True, monkeypatch)
    _same(_run(m0, c0, 17), _run(m1, c1, 17))
    ids = torch.randint(3, 32000, (1, 3)).to(DEV)
    monkeypatch.setattr(ops, "DRAFT_PERSIST", False)
    want = m0.forward(ids, c0, c0, 2, probs=(0.6, 0.9))
    monkeypatch.setattr(ops, "DRAFT_PERSIST", True)
    mirror = m1._persist.enable_mirror()
    old_ms = L.tf_draft_persist_tune(0, 50)
    try:
        for edge in (0, 3, 6, 10, 11):                                 # qkv[0], gate|up[0], attention[1], lm, top-p
            L.tf_draft_persist_tune(1, edge + 1)
            bad = m1.forward(ids, c1, c1, 2, probs=(0.6, 0.9))
            torch.cuda.synchronize()
            L.tf_draft_persist_tune(1, 0)
            err = L.tf_draft_persist_error(ops._ptr(m1._persist.ctl))
            assert err != 0 and int(mirror[0]) == err, (edge, err)
            assert bool(torch.isnan(bad.logits).any()) and bool(torch.isnan(bad.probs).any())
            again = m1.forward(ids, c1, c1, 2, probs=(0.6, 0.9))      # sticky: nothing runs on a failed block
            torch.cuda.synchronize()
            assert bool(torch.isnan(again.logits).all()) and bool(torch.isnan(again.probs).all())
            m1._persist.reset()
            assert m1._persist.error() == 0
            good = m1.forward(ids, c1, c1, 2, probs=(0.6, 0.9))
            torch.cuda.synchronize()
            assert torch.equal(good.logits, want.logits) and torch.equal(good.probs, want.probs), edge
    finally:
        L.tf_draft_persist_tune(1, 0)
        L.tf_draft_persist_tune(0, old_ms)
