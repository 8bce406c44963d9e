"""Round 4: the k-octet-major activation layout of the decode path (include/triforce_hip.h tf_skinny_gemm_act,
triforce_amd.ops.Act) on a real MI355X.  The layout changes only WHERE an element lives, never how a dot product is
accumulated, so every op must give bit-identical values in both layouts; the two-panels-per-wave launch form (4 waves)
must equal the one-panel form bit for bit as well.  Oracle parity of the arithmetic itself is covered by the row-major
tests (tests/test_gpu_ops.py) these results are compared with.  Reference call sites: models/modeling_llama.py:156-159,
212-245,278-284,408; models/tensor_op.py:52-64,140-181,276-360."""
import pytest
import torch

from tests import helpers as Hh

from oracle import ref_ops as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ops():
    from triforce_amd import ops
    return ops


def rnd(*shape, seed=0, scale=1.0, dtype=torch.float16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def test_pack_unpack_is_a_permutation_and_embed_rows_matches_index():
    ops = _ops()
    x = rnd(17, 256, seed=1).to(DEV)
    for R_ in (None, 17, 32):
        xp = ops.pack_act(x, R_)
        assert torch.equal(ops.unpack_act(xp, 17), x)
        a = ops.Act(xp, 17)
        assert a.shape == (17, 256) and a.sm == 8 and a.sk == 8 * xp.shape[1] and torch.equal(a.rows(), x)
    emb = rnd(1000, 512, seed=2).to(DEV)
    for n in (1, 7, 17, 32):
        ids = torch.randint(0, 1000, (1, n), generator=torch.Generator().manual_seed(n)).to(DEV)
        got = ops.embed_rows(emb, ids, True)
        assert isinstance(got, ops.Act) and torch.equal(got.rows(), emb[ids.reshape(-1)])
        assert torch.equal(ops.embed_rows(emb, ids, False), emb[ids.reshape(-1)])
        into = ops.Act.empty(n, 512, DEV)
        assert ops.embed_rows(emb, ids, True, out=into) is into and torch.equal(into.rows(), emb[ids.reshape(-1)])


@pytest.mark.parametrize("M", [1, 7, 8, 16, 17, 18, 31, 32])
@pytest.mark.parametrize("N,K", [(768, 768), (4096, 4096), (4096, 1376), (15360, 5120), (32000, 768)])
def test_gemm_layouts_are_bit_identical(M, N, K):
    """Plain, residual + sum-of-squares, norm prologue (re-reading and folding the hand-off), fp32 logits: k-octet-major
    operands against row-major ones, same kernel.  (15360, 5120) is the 13B q|k|v width: 480 panel pairs, so the
    two-panels-per-wave form runs — checked against the one-panel form too."""
    ops = _ops()
    from triforce_amd import hip
    eps = 1e-5
    w = rnd(N, K, seed=300 + M, scale=0.05).to(DEV)
    x, res = rnd(M, K, seed=301).to(DEV), rnd(M, N, seed=302).to(DEV)
    ln = (1 + 0.1 * rnd(K, seed=303).float()).half().to(DEV)
    pl = ops.PackedLinear(w)
    xa = ops.Act.from_rows(x)
    # plain
    y = ops.linear(x, pl)
    ya = ops.linear(xa, pl)
    assert isinstance(ya, ops.Act) and torch.equal(ya.rows(), y)
    # mixed: packed in, row-major out and the reverse
    assert torch.equal(ops.linear(xa, pl, out=torch.empty(M, N, dtype=torch.float16, device=DEV)), y)
    assert torch.equal(ops.linear(x, pl, out=ops.Act.empty(M, N, DEV)).rows(), y)
    # the two-panels-per-wave launch form is the same arithmetic
    L = hip.lib()
    old = L.tf_sg_tune(0, 33)
    try:
        assert torch.equal(ops.linear(x, pl), y) and torch.equal(ops.linear(xa, pl).rows(), y)
    finally:
        L.tf_sg_tune(0, old)
    # residual in place + sum-of-squares hand-off
    b1, b2 = res.clone(), ops.Act.from_rows(res)
    s1, s2 = ops.ss_buffer(N, DEV), ops.ss_buffer(N, DEV)
    ops.linear(x, pl, ln=ln, eps=eps, resid=b1, out=b1, ss_out=s1)
    ops.linear(xa, pl, ln=ln, eps=eps, resid=b2, out=b2, ss_out=s2)
    assert torch.equal(b2.rows(), b1) and torch.equal(s1[:, :M], s2[:, :M])
    # norm prologue folding the hand-off, and fp32 logits
    if N % 32 == 0 and N <= 15360:
        w2 = ops.PackedLinear(rnd(256, N, seed=304, scale=0.05).to(DEV))
        ln2 = (1 + 0.1 * rnd(N, seed=305).float()).half().to(DEV)
        assert torch.equal(ops.linear(b2, w2, ln=ln2, eps=eps, ss_in=s2).rows(), ops.linear(b1, w2, ln=ln2, eps=eps, ss_in=s1))
        assert torch.equal(ops.linear(b2, w2, ln=ln2, eps=eps, out_f32=True), ops.linear(b1, w2, ln=ln2, eps=eps, out_f32=True))
    # R > M: rows beyond M are never read or written
    xr = ops.Act.from_rows(x, R=32)
    yr = ops.Act(torch.full((N // 8, 32, 8), 7.0, dtype=torch.float16, device=DEV), M)
    ops.linear(xr, pl, out=yr)
    assert torch.equal(yr.rows(), y) and bool((yr.t[:, M:] == 7.0).all())


@pytest.mark.parametrize("M,I,K", [(1, 3072, 768), (7, 11008, 4096), (17, 13824, 5120), (18, 1728, 5120), (32, 768, 256)])
def test_swiglu_layouts_are_bit_identical(M, I, K):
    ops = _ops()
    from triforce_amd import hip
    x, wgu = rnd(M, K, seed=320 + M).to(DEV), rnd(2 * I, K, seed=321, scale=0.05).to(DEV)
    ln = (1 + 0.1 * rnd(K, seed=322).float()).half().to(DEV)
    pl = ops.PackedLinear(wgu, split=2)
    want = ops.mlp_act(x, pl)
    got = ops.mlp_act(ops.Act.from_rows(x), pl)
    assert isinstance(got, ops.Act) and torch.equal(got.rows(), want)
    assert torch.equal(ops.mlp_act(ops.Act.from_rows(x), pl, ln=ln, eps=1e-5).rows(), ops.mlp_act(x, pl, ln=ln, eps=1e-5))
    L = hip.lib()
    old = L.tf_sg_tune(0, 33)
    try:
        assert torch.equal(ops.mlp_act(x, pl), want)            # (17, 13824, 5120): 432 panel pairs -> two per wave above
    finally:
        L.tf_sg_tune(0, old)
    # against the oracle at the new launch form (absolute bound of the row-major test)
    gu = R.linear(x.cpu(), wgu.cpu())
    ref = R.silu_mul(gu[:, :I], gu[:, I:])
    d = (got.rows().float().cpu() - ref.float()).abs()
    bound = 2.0 ** -9 * (1.0 + gu[:, :I].float().abs()) * (1.0 + gu[:, I:].float().abs())
    assert (d <= bound).all()


@pytest.mark.parametrize("M,H,D,K", [(7, 32, 128, 4096), (17, 40, 128, 5120), (18, 4, 64, 768), (32, 3, 128, 512)])
def test_qkv_rope_layouts_are_bit_identical(M, H, D, K):
    ops = _ops()
    T, slot0, eps = 64, 5, 1e-5
    w = rnd(3 * H * D, K, seed=330 + M, scale=0.05).to(DEV)
    x = rnd(M, K, seed=331).to(DEV)
    ln = (1 + 0.1 * rnd(K, seed=332).float()).half().to(DEV)
    cos, sin = R.rope_tables_yarn(D, 4096, 16.0, 256) if D == 128 else R.rope_tables_plain(D, 4096)
    pos = torch.randint(0, 4096, (M,), generator=torch.Generator().manual_seed(M)).to(DEV)
    pl = ops.PackedLinear(w, rope=(H, D))
    outs = []
    for xin in (x, ops.Act.from_rows(x)):
        k = torch.zeros(H, T, D, dtype=torch.float16, device=DEV)
        v = torch.zeros(H, T, D, dtype=torch.float16, device=DEV)
        q = ops.qkv_rope(xin, pl, ln, eps, cos.to(DEV), sin.to(DEV), pos, k, v, slot0, H, D)
        outs.append((q, k, v))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("sq,sk,H,D,nsplit", [(7, 4103, 32, 128, None), (17, 12305, 5, 128, None), (18, 3000, 40, 128, None),
                                              (8, 777, 4, 64, 3), (32, 600, 2, 128, 2)])
def test_attn_decode_packed_output_is_the_same_values(sq, sk, H, D, nsplit, monkeypatch):
    """Output of the split-KV attention written k-octet-major (what o_proj then reads), one-launch and two-launch merge."""
    ops = _ops()
    q = rnd(sq, H, D, seed=340).to(DEV)
    k, v = rnd(H, sk + 8, D, seed=341).to(DEV), rnd(H, sk + 8, D, seed=342).to(DEV)
    want = ops.attn_decode(q, k, v, sk, 0.0883, nsplit=nsplit)
    got = ops.attn_decode(q, k, v, sk, 0.0883, nsplit=nsplit, packed=True)
    assert isinstance(got, ops.Act) and got.shape == (sq, H * D) and torch.equal(got.rows(), want)
    monkeypatch.setattr(ops, "ATTN_FUSED_MERGE", False)
    assert torch.equal(ops.attn_decode(q, k, v, sk, 0.0883, nsplit=nsplit, packed=True).rows(), want)


@pytest.mark.parametrize("rows,hidden,world", [(7, 4096, 2), (17, 5120, 2), (18, 768, 4), (32, 4096, 2)])
@pytest.mark.parametrize("alternate", [False, True])
def test_exchange_with_packed_residual_stream(rows, hidden, world, alternate):
    """tf_allreduce_oneshot_act (virtual ranks on one device): sum + residual bit-identical to the row-major exchange,
    sums of squares per (panel, row) equal to fp32 rounding of a different summation order."""
    ops = _ops()
    from triforce_amd.utils.oneshot_ar import OneShotAllReduce, reference_sum
    dev = torch.device(DEV)
    group = OneShotAllReduce.local_group(world, dev, 32 * hidden, alternate=alternate)
    streams = _rank_streams(world)
    try:
        for rep in range(4):                                   # even count: the alternating form stays in step
            parts = [rnd(rows, hidden, seed=350 + 10 * rep + r).to(dev) for r in range(world)]
            resid = rnd(rows, hidden, seed=349 + rep).to(dev)
            want = reference_sum(parts, resid)
            outs, sss = [], []
            torch.cuda.synchronize()
            for r, (g, s) in enumerate(zip(group, streams)):
                with torch.cuda.stream(s):
                    st = g.staging(rows, hidden, packed=True)
                    st.copy_(parts[r])
                    x = ops.Act.from_rows(resid)
                    ss = ops.ss_buffer(hidden, dev)
                    g.reduce(st, x, resid=x, ss_out=ss)
                    outs.append(x)
                    sss.append(ss)
            torch.cuda.synchronize()
            for g in group:
                assert g.error() == 0 and g.error_device() == 0
            for x, ss in zip(outs, sss):
                assert torch.equal(x.rows(), want)
                Hh.close(ss[:, :rows], (want.float() ** 2).view(rows, hidden // 16, 16).sum(-1).t(),
                                           rtol=2e-6, atol=1e-6)
    finally:
        for g in group:
            g.close()


def _rank_streams(world):
    """The process-wide rank streams of tests/test_gpu_tp_offload.py (virtual ranks need distinct hardware queues)."""
    from tests.test_gpu_tp_offload import _rank_streams as shared
    return shared(world)


@pytest.mark.parametrize("q_len", [7, 17])
def test_model_forward_is_bit_identical_in_both_layouts(q_len, monkeypatch):
    """The whole fused decode forward (retrieval-cache and full-cache forms) with k-octet-major activations against the
    row-major build of the same layer: identical logits, identical KV rows."""
    ops = _ops()
    from oracle import specs
    from triforce_amd.models.cache import FlashSimpleCache
    from triforce_amd.models.config_yarn import LlamaConfig
    from triforce_amd.models.modeling_llama import LlamaForCausalLM
    cfg = LlamaConfig.from_dict(specs.tiny_target_config(vocab_size=2048))
    model = LlamaForCausalLM(cfg, DEV).init_random(3)
    ids = torch.randint(0, cfg.vocab_size, (1, 40 + q_len), generator=torch.Generator().manual_seed(5)).to(DEV)
    res = {}
    for layout in ("rows", "packed"):
        monkeypatch.setattr(ops, "ACT_LAYOUT", layout)
        cache = FlashSimpleCache(model, 256)
        model(input_ids=ids[:, :40], kv_cache=cache)                       # prefill block (hipBLASLt path)
        logits = model(input_ids=ids[:, 40:], kv_cache=cache).logits
        res[layout] = (logits.clone(), cache.k.clone(), cache.v.clone())
    for a, b in zip(res["rows"], res["packed"]):
        assert torch.equal(a, b)
