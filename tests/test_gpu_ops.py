"""Op-level parity on a real MI355X: every HIP kernel, called through the C ABI (triforce_amd.ops ->
ctypes -> libtriforce_hip.so), against its CPU oracle on the same seeded inputs.

Bars: bit-exact for integer / index / copy / fp16-elementwise work; for reductions in fp32 whose
summation order legitimately differs from the CPU's the tolerance is written next to the assert.
"""
import math

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


def ulp_report(name, got, want, max_ulp_frac=2e-3, atol=0.0, ulps=1):
    """fp16 results that may differ by rounding of an fp32 reduction: at most `max_ulp_frac` of the
    elements may differ at all, and none by more than `ulps` fp16 ulp (+atol)."""
    got, want = got.float().cpu(), want.float().cpu()
    diff = (got - want).abs()
    ulp = torch.maximum(want.abs(), torch.tensor(6.1e-5)) * 2 ** -10
    bad = diff > (ulp * (ulps + 0.01) + atol)
    frac = (diff > 0).float().mean().item()
    Hh.record("ulp_report", float((diff / (ulp * (ulps + 0.01) + atol)).max()), what=name, max_abs=float(diff.max()),
              mean_abs=float(diff.mean()), ulps=ulps, atol=float(atol), frac_differing=frac, frac_allowed=max_ulp_frac,
              frac_used=frac / max_ulp_frac, max_ref=float(want.abs().max()))
    assert not bad.any(), f"{name}: {int(bad.sum())} elements off by >{ulps} ulp, max diff {diff.max().item():.3e}"
    assert frac <= max_ulp_frac, f"{name}: {frac:.4%} elements differ (allowed {max_ulp_frac:.2%})"


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,hidden", [(1, 256), (7, 768), (128, 4096), (3, 5120)])
@pytest.mark.parametrize("residual", [False, True])
def test_rmsnorm(rows, hidden, residual):
    ops = _ops()
    x, w = rnd(rows, hidden, seed=1), (1 + 0.1 * rnd(hidden, seed=2).float()).half()
    res = rnd(rows, hidden, seed=3) if residual else None
    xs = x + res if residual else x
    want = R.rms_norm(xs, w, 1e-5)
    xd = x.to(DEV)
    so = torch.empty_like(xd) if residual else None
    got = ops.rmsnorm(xd, w.to(DEV), 1e-5, residual=res.to(DEV) if residual else None, sum_out=so)
    if residual:
        assert torch.equal(so.cpu(), xs), "residual sum must be bit-exact (fp16 add)"
    # fp32 sum of squares is reduced in a different order than torch CPU: the normalised value may round
    # to the neighbouring fp16 for a handful of elements, and the following fp16 weight multiply (|w|~1.1)
    # can carry that to 2 ulp of the product -> <=2 ulp on <=1% of the elements
    ulp_report("rmsnorm", got, want, max_ulp_frac=1e-2, ulps=2)


def test_rmsnorm_inplace_residual_accumulate():
    ops = _ops()
    x, d, w = rnd(8, 512, seed=4).to(DEV), rnd(8, 512, seed=5).to(DEV), torch.ones(512, dtype=torch.float16, device=DEV)
    want_sum = (x.cpu() + d.cpu())
    ops.rmsnorm(d, w, 1e-6, residual=x, sum_out=x)
    assert torch.equal(x.cpu(), want_sum)


@pytest.mark.parametrize("H,D", [(4, 64), (2, 128), (32, 128)])
@pytest.mark.parametrize("rotate_k", [True, False])
@pytest.mark.parametrize("head_major", [True, False])
def test_rope_append_bit_exact(H, D, rotate_k, head_major):
    ops = _ops()
    rows, T, slot0 = 9, 64, 17
    qkv = rnd(rows, 3 * H * D, seed=6)
    cos, sin = R.rope_tables_yarn(D, 4096, 16.0, 256) if D == 128 else R.rope_tables_plain(D, 4096)
    pos = torch.tensor([4000, 0, 1, 77, 2048, 5, 6, 7, 4095])
    q = qkv[:, :H * D].view(rows, H, D)
    k = qkv[:, H * D:2 * H * D].view(rows, H, D)
    v = qkv[:, 2 * H * D:].view(rows, H, D)
    want_q = R.apply_rope(q, cos, sin, pos)
    want_k = R.apply_rope(k, cos, sin, pos) if rotate_k else k
    if head_major:
        kc = torch.zeros(H, T, D, dtype=torch.float16, device=DEV)
        vc = torch.zeros(H, T, D, dtype=torch.float16, device=DEV)
        kl, vl = kc, vc
    else:
        kc = torch.zeros(T, H, D, dtype=torch.float16, device=DEV)
        vc = torch.zeros(T, H, D, dtype=torch.float16, device=DEV)
        kl, vl = kc.permute(1, 0, 2), vc.permute(1, 0, 2)
    got_q = ops.rope_append(qkv.to(DEV), cos.to(DEV), sin.to(DEV), pos.to(DEV), kl, vl, slot0, H, D, rotate_k=rotate_k)
    assert torch.equal(got_q.cpu(), want_q)
    assert torch.equal(kl[:, slot0:slot0 + rows].permute(1, 0, 2).cpu(), want_k)
    assert torch.equal(vl[:, slot0:slot0 + rows].permute(1, 0, 2).cpu(), v)
    assert kl[:, :slot0].abs().sum() == 0 and kl[:, slot0 + rows:].abs().sum() == 0
    # device-side slot (graph-capturable form)
    kl.zero_()
    sdev = torch.tensor([3], dtype=torch.int32, device=DEV)
    ops.rope_append(qkv.to(DEV), cos.to(DEV), sin.to(DEV), pos.to(DEV), kl, vl, 0, H, D, rotate_k=rotate_k, slot0_dev=sdev)
    assert torch.equal(kl[:, 3:3 + rows].permute(1, 0, 2).cpu(), want_k)


@pytest.mark.parametrize("rows,I", [(1, 768), (7, 11008), (128, 3072)])
def test_silu_mul(rows, I):
    ops = _ops()
    gu = rnd(rows, 2 * I, seed=7, scale=2.0)
    want = R.silu_mul(gu[:, :I], gu[:, I:])
    got = ops.silu_mul(gu.to(DEV))
    # expf on the device vs the CPU's vectorised exp: <=1 ulp on a small fraction
    ulp_report("silu_mul", got, want, max_ulp_frac=2e-2)


# ------------------------------------------------------------------------------------------
ATTN_CASES = [
    # sq, sk, H, D, nsplit
    (1, 1, 2, 128, None), (1, 17, 2, 128, None), (7, 16, 4, 128, 1), (7, 261, 4, 128, None), (8, 300, 2, 64, 2),
    (5, 1000, 4, 64, 3), (16, 4103, 4, 128, None), (17, 12305, 2, 128, None), (18, 2049, 4, 128, 5),
    (32, 333, 2, 128, None), (32, 32, 2, 64, None), (19, 4096, 4, 64, 7), (7, 4103, 32, 128, None),
]


def _attn_inputs(sq, sk, H, D, seed, head_major=True, cap=None):
    cap = cap or sk
    q, k, v = rnd(sq, H, D, seed=seed), rnd(cap, H, D, seed=seed + 1), rnd(cap, H, D, seed=seed + 2)
    if head_major:
        kd, vd = k.permute(1, 0, 2).contiguous().to(DEV), v.permute(1, 0, 2).contiguous().to(DEV)
    else:
        kd, vd = k.to(DEV).permute(1, 0, 2), v.to(DEV).permute(1, 0, 2)
    return q, k, v, kd, vd


# attention tolerance of the BLOCK kernels (prefill chunks, tree verify): P is rounded ONCE to fp16 before the PV MFMA
# (like flash-attn) and exp is the fast hardware exp2 path; the oracle keeps P in fp32.  |out| <= max|v| ~ 4, so
# 2e-3 abs + 2e-3 rel.
ATTN_ATOL, ATTN_RTOL = 2e-3, 2e-3
# ... and of the split-KV decode / verify kernel, which feeds P to the matrix core as hi + lo fp16 parts (round 3):
# measured against attention accumulated in fp64, 0.16 % of its outputs differ from the exactly rounded value (mean
# error 0.25 fp16 ulp, profiles/r03_attn_pipeline_ab.jsonl) and against the fp32 CPU oracle max |d| <= 3.1e-5 on
# outputs of magnitude ~0.02 (profiles/r03*_parity_notes.txt).  One fp16 ulp is 2^-10 relative at most: 1.5e-3 rel
# covers a one-ulp disagreement between two correctly-rounded-ish results, 2e-5 abs the outputs near zero.
DECODE_ATOL, DECODE_RTOL = 2e-5, 1.5e-3


@pytest.mark.parametrize("sq,sk,H,D,nsplit", ATTN_CASES)
def test_attn_decode_matches_oracle(sq, sk, H, D, nsplit):
    ops = _ops()
    scale = R.softmax_scale_for(D)
    q, k, v, kd, vd = _attn_inputs(sq, sk, H, D, seed=10 + sq + sk)
    want = R.attn_kvcache(q, k, v, scale).reshape(sq, H * D)
    got = ops.attn_decode(q.to(DEV), kd, vd, sk, scale, nsplit=nsplit)
    Hh.close(got.float().cpu(), want.float(), atol=DECODE_ATOL, rtol=DECODE_RTOL)


@pytest.mark.parametrize("sq,sk,H,D,nsplit", ATTN_CASES + [(7, 4103, 32, 128, 8), (17, 12305, 16, 128, 48),
                                                         (1, 130000, 4, 128, 64), (6, 300, 12, 64, None),
                                                         # round 4: many splits merged INSIDE the launch on small grids
                                                         # (the 4 / 5 heads of a tensor-parallel rank), odd counts too
                                                         (7, 4103, 4, 128, 32), (17, 12305, 5, 128, 51),
                                                         (18, 20000, 5, 128, 51), (8, 5000, 8, 128, 33),
                                                         (32, 3000, 4, 128, 64), (7, 4103, 4, 128, 9)])
def test_attn_decode_one_launch_merge_is_bit_identical_to_two_launches(sq, sk, H, D, nsplit, monkeypatch):
    """tf_attn_decode_fused (the last workgroup of a head merges its splits inside the split kernel) against
    tf_attn_decode (split kernel + merge kernel): same arithmetic in the same order -> the same bits; repeated
    launches on one ticket row (the row must come back zero each time), and replayed from a hipGraph."""
    ops = _ops()
    from triforce_amd import hip
    # > 8 splits merge inside the launch only with the rendezvous form switched on (shipped off: it measured slower,
    # profiles/r04_attn_rendezvous_merge_ab.jsonl) — on here so that the path stays checked
    was = hip.lib().tf_attn_tune(0, 1)
    try:
        _one_launch_merge_case(ops, sq, sk, H, D, nsplit, monkeypatch)
    finally:
        hip.lib().tf_attn_tune(0, was)


def _one_launch_merge_case(ops, sq, sk, H, D, nsplit, monkeypatch):
    scale = R.softmax_scale_for(D)
    q, k, v, kd, vd = _attn_inputs(sq, sk, H, D, seed=31 + sq + sk)
    qd = q.to(DEV)
    monkeypatch.setattr(ops, "ATTN_FUSED_MERGE", False)
    two = ops.attn_decode(qd, kd, vd, sk, scale, nsplit=nsplit)
    monkeypatch.setattr(ops, "ATTN_FUSED_MERGE", True)
    for _ in range(3):
        one = ops.attn_decode(qd, kd, vd, sk, scale, nsplit=nsplit)
        assert torch.equal(one, two)
    row = ops._ticket_row(qd.device, torch.cuda.current_stream().cuda_stream or 0)
    torch.cuda.synchronize()
    assert int(row.abs().sum()) == 0
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.attn_decode(qd, kd, vd, sk, scale, nsplit=nsplit)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = ops.attn_decode(qd, kd, vd, sk, scale, nsplit=nsplit)
    for _ in range(3):
        captured.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(captured, two)


def test_attn_decode_token_major_and_device_seqlen():
    ops = _ops()
    sq, sk, H, D, cap = 7, 777, 4, 128, 1024
    scale = R.softmax_scale_for(D)
    q, k, v, kd, vd = _attn_inputs(sq, sk, H, D, seed=99, head_major=False, cap=cap)
    want = R.attn_kvcache(q, k[:sk], v[:sk], scale).reshape(sq, H * D)
    got = ops.attn_decode(q.to(DEV), kd, vd, sk, scale)
    Hh.close(got.float().cpu(), want.float(), atol=DECODE_ATOL, rtol=DECODE_RTOL)
    # key count read from device memory, launch sized by the capacity
    skd = torch.tensor([sk], dtype=torch.int32, device=DEV)
    got2 = ops.attn_decode(q.to(DEV), kd, vd, cap, scale, sk_dev=skd, nsplit=4)
    Hh.close(got2.float().cpu(), want.float(), atol=DECODE_ATOL, rtol=DECODE_RTOL)


def test_attn_decode_online_softmax_rescale_is_exercised():
    """One key far above the rest, placed late in the stream of one split and early in another, forces
    the running-max rescale branch (a rare data-dependent path needs its own test)."""
    ops = _ops()
    sq, sk, H, D = 8, 2000, 2, 128
    scale = R.softmax_scale_for(D)
    q, k, v, _, _ = _attn_inputs(sq, sk, H, D, seed=5)
    k[1500] = q[3] * 3.0            # huge score for query 3 at key 1500
    k[40, 1] = q[6, 1] * 2.5
    kd, vd = k.permute(1, 0, 2).contiguous().to(DEV), v.permute(1, 0, 2).contiguous().to(DEV)
    want = R.attn_kvcache(q, k, v, scale).reshape(sq, H * D)
    for ns in (1, 2, 5):
        got = ops.attn_decode(q.to(DEV), kd, vd, sk, scale, nsplit=ns)
        Hh.close(got.float().cpu(), want.float(), atol=DECODE_ATOL, rtol=DECODE_RTOL)


@pytest.mark.parametrize("sq,sk,H,D", [(128, 640, 2, 128), (128, 128, 2, 128), (64, 64, 3, 64), (100, 1000, 2, 128),
                                        (33, 5000, 4, 128), (128, 9000, 4, 64), (300, 700, 2, 128)])
def test_attn_prefill_block_equals_oracle(sq, sk, H, D):
    """Chunked-prefill attention (tf_attn_block: <=128 rows per pass; 300 rows = three bottom-right-aligned slabs)."""
    ops = _ops()
    scale = R.softmax_scale_for(D)
    q, k, v, kd, vd = _attn_inputs(sq, sk, H, D, seed=3 + sq)
    want = R.attn_kvcache(q, k, v, scale).reshape(sq, H * D)
    got = ops.attn_prefill(q.to(DEV), kd, vd, sk, scale)
    Hh.close(got.float().cpu(), want.float(), atol=ATTN_ATOL, rtol=ATTN_RTOL)


@pytest.mark.parametrize("sq,sk,H,D", [(300, 700, 2, 128), (1024, 1024, 4, 128), (1024, 5000, 8, 128), (1000, 3333, 4, 64),
                                        (129, 129, 2, 128), (2048, 4096, 4, 128), (640, 70000, 8, 128)])
def test_attn_prefill_one_launch_equals_oracle_and_block_by_block(sq, sk, H, D, monkeypatch):
    """tf_attn_prefill (every 128-row block of the chunk in one launch, common key partition, neutral partials past a
    block's causal edge) against the oracle and against one tf_attn_block launch per block; ragged last block, chunks
    that ARE the whole cache (sk == sq), more than 8 row blocks, and causality (perturbing the last key moves only the
    last row)."""
    ops = _ops()
    scale = R.softmax_scale_for(D)
    q, k, v, kd, vd = _attn_inputs(sq, sk, H, D, seed=41 + sq)
    qd = q.to(DEV)
    monkeypatch.setattr(ops, "ATTN_PREFILL_ONE_LAUNCH", True)
    got = ops.attn_prefill(qd, kd, vd, sk, scale)
    monkeypatch.setattr(ops, "ATTN_PREFILL_ONE_LAUNCH", False)
    blocks = ops.attn_prefill(qd, kd, vd, sk, scale)
    Hh.close(got.float(), blocks.float(), atol=1e-3, rtol=ATTN_RTOL)
    if sq * sk <= 1024 * 5000:                       # the CPU oracle materialises sq x sk scores per head
        want = R.attn_kvcache(q, k, v, scale).reshape(sq, H * D)
        Hh.close(got.float().cpu(), want.float(), atol=ATTN_ATOL, rtol=ATTN_RTOL)
    monkeypatch.setattr(ops, "ATTN_PREFILL_ONE_LAUNCH", True)
    kd2 = kd.clone()
    kd2[:, sk - 1] += 1.0
    moved = (ops.attn_prefill(qd, kd2, vd, sk, scale) != got).any(dim=1)
    assert not moved[:-1].any()


def test_attn_block_splits_agree_and_match_decode_kernel():
    """Same block through 1 / 3 / 16 KV splits and, row slab by row slab, through the <=32-row decode kernel."""
    ops = _ops()
    sq, sk, H, D = 96, 3000, 2, 128
    scale = R.softmax_scale_for(D)
    q, k, v, kd, vd = _attn_inputs(sq, sk, H, D, seed=77)
    qd = q.to(DEV)
    ref = torch.cat([ops.attn_decode(qd[r:r + 32].contiguous(), kd, vd, sk - (sq - r - 32), scale) for r in (0, 32, 64)])
    for ns in (1, 3, 16):
        got = ops.attn_block(qd, kd, vd, sk, scale, nsplit=ns)
        Hh.close(got.float(), ref.float(), atol=5e-4, rtol=ATTN_RTOL)


def test_attn_block_full_size_prefill_chunk():
    """A 128-token prefill chunk at the END of a BASELINE configs[1] prompt (124 928 keys, 32 heads): the one-pass block
    kernel must agree with the <=32-row decode kernel run slab by slab (itself checked against the oracle at this
    size below), and be causal — perturbing the last key changes only the last query row."""
    ops = _ops()
    sq, sk, H, D = 128, 124928, 32, 128
    scale = R.softmax_scale_for(D)
    g = torch.Generator(device=DEV).manual_seed(11)
    kd = torch.randn(H, sk, D, generator=g, device=DEV, dtype=torch.float16)
    vd = torch.randn(H, sk, D, generator=g, device=DEV, dtype=torch.float16)
    qd = torch.randn(sq, H, D, generator=g, device=DEV, dtype=torch.float16)
    got = ops.attn_block(qd, kd, vd, sk, scale)
    ref = torch.cat([ops.attn_decode(qd[r:r + 32].contiguous(), kd, vd, sk - (sq - r - 32), scale) for r in range(0, sq, 32)])
    Hh.close(got.float(), ref.float(), atol=6e-5, rtol=ATTN_RTOL)
    kd[:, sk - 1] += 4.0
    got2 = ops.attn_block(qd, kd, vd, sk, scale)
    assert torch.equal(got2[:sq - 1], got[:sq - 1]) and not torch.equal(got2[sq - 1], got[sq - 1])


def test_attn_decode_full_size_cfg2_layer():
    """BASELINE configs[1] shape for one layer: 8 queries x 124 935 keys x 32 heads x 128 (2 GB of KV)."""
    ops = _ops()
    sq, sk, H, D = 8, 124928 + 7, 32, 128
    scale = R.softmax_scale_for(D)
    g = torch.Generator(device=DEV).manual_seed(1)
    kd = torch.randn(H, sk, D, generator=g, device=DEV, dtype=torch.float16)
    vd = torch.randn(H, sk, D, generator=g, device=DEV, dtype=torch.float16)
    q = torch.randn(sq, H, D, generator=g, device=DEV, dtype=torch.float16)
    got = ops.attn_decode(q, kd, vd, sk, scale)
    want = R.attn_kvcache(q.cpu(), kd.permute(1, 0, 2).cpu(), vd.permute(1, 0, 2).cpu(), scale).reshape(sq, H * D)
    Hh.close(got.float().cpu(), want.float(), atol=1e-5, rtol=DECODE_RTOL)
    # size-independent property: attention is linear in V (doubling V is exact in fp16: the same bits, one exponent up)
    got2 = ops.attn_decode(q, kd, vd * 2, sk, scale)
    Hh.close(got2.float(), got.float() * 2, atol=2 * DECODE_ATOL, rtol=DECODE_RTOL)


@pytest.mark.parametrize("sq,kv_len,H", [(1, 5, 2), (3, 64, 12), (9, 259, 12), (64, 200, 12), (7, 130, 3), (64, 314, 12), (128, 378, 12), (17, 384, 4),
                                          (5, 400, 2)])
def test_attn_rope_on_read(sq, kv_len, H):
    ops = _ops()
    D = 64
    scale = R.softmax_scale_for(D)
    cos, sin = R.rope_tables_plain(D, 2048)
    q, k, v, kd, vd = _attn_inputs(sq, kv_len, H, D, seed=20 + sq, cap=kv_len + 3)
    kr = R.apply_rope(k[:kv_len], cos, sin, torch.arange(kv_len))
    want = R.attn_kvcache(q, kr, v[:kv_len], scale).reshape(sq, H * D)
    got = ops.attn_rope_on_read(q.to(DEV), kd, vd, cos.to(DEV), sin.to(DEV), kv_len, scale)
    Hh.close(got.float().cpu(), want.float(), atol=6e-4, rtol=ATTN_RTOL)


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("T,H,D,chunk", [(1000, 4, 64, 8), (2048, 2, 128, 8), (124928, 32, 128, 8), (640, 3, 128, 16)])
def test_retrieval_score(T, H, D, chunk):
    ops = _ops()
    C = T // chunk
    k, q = rnd(T + 5, H, D, seed=30), rnd(H, D, seed=31)
    want = R.retrieval_scores(k, q, C * chunk, chunk)
    kd = k.permute(1, 0, 2).contiguous().to(DEV)
    got = ops.retrieval_score(kd, q.to(DEV), C, chunk)
    # two fp16 rounding points (mean, dot) with fp32 sums in a different order than torch CPU:
    # <= 1 ulp (SURVEY §7 "scores (<=1 ulp fp16)"); most elements identical
    ulp_report("retrieval_score", got, want, max_ulp_frac=0.05, atol=2e-3)


@pytest.mark.parametrize("H,C,sets", [(1, 9, 4), (4, 125, 16), (2, 256, 32), (32, 15616, 512), (16, 16256, 1536), (3, 3000, 2999)])
def test_retrieval_topk_bit_exact_with_ties(H, C, sets):
    ops = _ops()
    # coarse quantisation -> thousands of exact ties, including at the k-th value
    s = (rnd(H, C, seed=40).float() * 4).round().div(4).half()
    s[0, 3] = float("inf")
    s[-1, C - 1] = float("-inf")
    want = R.retrieval_topk(s, sets)
    got = ops.retrieval_topk(s.to(DEV), sets)
    assert got.dtype == torch.int32
    assert torch.equal(got.cpu().long(), want), "top-k indices (descending score, ties -> lowest chunk) must be bit-exact"


@pytest.mark.parametrize("H,D,chunk,C,sets", [(4, 64, 8, 125, 16), (32, 128, 8, 1000, 512), (2, 128, 16, 40, 8),
                                              (3, 128, 8, 60, 13), (5, 64, 4, 33, 1)])      # ragged slot groups
def test_retrieval_gather_bit_exact(H, D, chunk, C, sets):
    ops = _ops()
    T = C * chunk
    k, v = rnd(T, H, D, seed=50), rnd(T, H, D, seed=51)
    g = torch.Generator().manual_seed(52)
    idx = torch.stack([torch.randperm(C, generator=g)[:sets] for _ in range(H)])
    want_k, want_v = R.retrieval_gather(k, idx, chunk), R.retrieval_gather(v, idx, chunk)
    kd, vd = k.permute(1, 0, 2).contiguous().to(DEV), v.permute(1, 0, 2).contiguous().to(DEV)
    R_ = sets * chunk + 7
    kr = torch.zeros(H, R_, D, dtype=torch.float16, device=DEV)
    vr = torch.zeros(H, R_, D, dtype=torch.float16, device=DEV)
    ops.retrieval_gather(kd, vd, idx.to(torch.int32).to(DEV), kr, vr, chunk)
    assert torch.equal(kr[:, :sets * chunk].permute(1, 0, 2).cpu(), want_k)
    assert torch.equal(vr[:, :sets * chunk].permute(1, 0, 2).cpu(), want_v)
    assert kr[:, sets * chunk:].abs().sum() == 0


def test_retrieval_build_round_trip_full_size():
    """cfg2 layer shape: score -> top-k -> gather; property checks that do not need the CPU oracle:
    chunk 0 first, indices unique, scores of selected chunks sorted descending, every selected score >=
    every unselected one, gathered rows equal the source rows."""
    ops = _ops()
    H, D, chunk, P, B = 32, 128, 8, 124928, 4096
    C, sets = P // chunk, B // chunk
    g = torch.Generator(device=DEV).manual_seed(7)
    kd = torch.randn(H, P, D, generator=g, device=DEV, dtype=torch.float16)
    vd = torch.randn(H, P, D, generator=g, device=DEV, dtype=torch.float16)
    q = torch.randn(H, D, generator=g, device=DEV, dtype=torch.float16)
    scores = ops.retrieval_score(kd, q, C, chunk)
    idx = ops.retrieval_topk(scores, sets)
    kr = torch.zeros(H, B + 7, D, dtype=torch.float16, device=DEV)
    vr = torch.zeros(H, B + 7, D, dtype=torch.float16, device=DEV)
    ops.retrieval_gather(kd, vd, idx, kr, vr, chunk)
    idxl = idx.long()
    assert (idxl[:, 0] == 0).all()
    assert all(len(set(r.tolist())) == sets for r in idxl.cpu())
    sel = torch.gather(scores.float(), 1, idxl[:, 1:])
    assert (sel[:, :-1] >= sel[:, 1:]).all(), "selected chunks must come in descending score order"
    mask = torch.ones_like(scores, dtype=torch.bool).scatter(1, idxl, False)
    mask[:, 0] = False
    unsel_max = scores.float().masked_fill(~mask, float("-inf")).max(dim=1).values
    assert (sel[:, -1] >= unsel_max).all()
    tok = (idxl.unsqueeze(-1) * chunk + torch.arange(chunk, device=DEV)).reshape(H, B)
    src = torch.gather(kd, 1, tok.unsqueeze(-1).expand(H, B, D))
    assert torch.equal(kr[:, :B], src)
    assert torch.equal(ops.retrieval_topk(scores, sets), idx)          # idempotent / deterministic
    # oracle on one head at full size (bit-exact top-k given identical scores)
    want = R.retrieval_topk(scores[:1].cpu(), sets)
    assert torch.equal(idxl[:1].cpu(), want)


def test_kv_copy_and_shift_rows():
    ops = _ops()
    L, H, D = 3, 4, 64
    src = rnd(L, H, 50, D, seed=60).to(DEV)
    dst = torch.zeros(L, H, 40, D, dtype=torch.float16, device=DEV)
    ops.kv_copy_rows(src, dst, 20, 7, 13)
    want = torch.zeros(L, H, 40, D, dtype=torch.float16)
    want[:, :, 7:20] = src.cpu()[:, :, 20:33]
    assert torch.equal(dst.cpu(), want)
    ops.kv_copy_rows(src[1:2], dst[1:2], 0, 0, 1)                       # layer slice views
    assert torch.equal(dst[1, :, 0].cpu(), src[1, :, 0].cpu())
    for n, shift in [(234, 5), (236, 1), (10, 30), (100, 0)]:
        c = rnd(2, 12, 300, D, seed=61 + n).to(DEV)
        ref = c.cpu().clone()
        ref[:, :, 16:16 + n] = ref[:, :, 16 + shift:16 + shift + n].clone()
        ops.kv_shift_rows(c, 16 + shift, 16, n)
        assert torch.equal(c.cpu(), ref), f"overlapping shift n={n} shift={shift}"


# ------------------------------------------------------------------------------------------
def _probs(rows, V, seed, sparse=False):
    lg = rnd(rows, V, seed=seed, dtype=torch.float32) * 3
    return R.norm_logits(lg, 0.6, -1, 0.9) if sparse else torch.softmax(lg, dim=-1)


@pytest.mark.parametrize("V", [512, 32000])
@pytest.mark.parametrize("sparse", [False, True])
def test_sample_inverse_cdf(V, sparse):
    ops = _ops()
    p = _probs(1, V, 70, sparse)[0]
    out = torch.zeros(1, dtype=torch.int64, device=DEV)
    for u in [0.0, 1e-7, 0.13, 0.5, 0.77, 0.999, 0.9999999]:
        ops.sample_inverse_cdf(p.to(DEV), torch.tensor([u], device=DEV), out)
        want = R.sample_inverse_cdf(p, u)
        got = int(out.item())
        assert p[got] > 0
        if got != want:        # only a cumulative-sum rounding tie may move the boundary by one non-zero bin
            c = torch.cumsum(p.double(), 0)
            assert abs(float(c[min(got, want)]) - u * float(c[-1])) < 1e-5, (u, got, want)
    onehot = torch.zeros(V)
    onehot[V - 3] = 1.0
    ops.sample_inverse_cdf(onehot.to(DEV), torch.tensor([0.42], device=DEV), out)
    assert int(out.item()) == V - 3


def test_accept_chain_and_middle_accept_match_oracle():
    ops = _ops()
    V = 32000
    out = torch.zeros(4, dtype=torch.int64, device=DEV)
    gen = torch.Generator().manual_seed(80)
    for trial in range(24):
        g2 = [1, 4, 6, 7, 17, 18][trial % 6]
        p = _probs(g2 + 1, V, 100 + trial, sparse=True)
        q = _probs(g2, V, 200 + trial, sparse=True)
        toks = torch.stack([torch.multinomial(q[i], 1, generator=gen)[0] for i in range(g2)])
        if trial % 3 == 0:                                    # make the drafted tokens likely under p too
            for i in range(g2):
                p[i, toks[i]] = p[i].max()
                p[i] /= p[i].sum()
        u = torch.rand(g2 + 1, generator=gen)
        eos = int(toks[min(1, g2 - 1)]) if trial % 4 == 0 else -1
        for inclusive in (False, True):
            want = R.accept_and_correct(p, q, toks.tolist(), u.tolist(), inclusive, eos)
            ops.accept_chain(p.to(DEV), q.to(DEV), toks.to(DEV), u.to(DEV), g2, inclusive, eos, out)
            got = tuple(out.tolist())
            assert got[0] == want[0] and got[2] == want[2] and got[3] == want[3], (trial, got, want)
            assert got[1] == want[1], (trial, got, want)
    # one inner step
    gamma = 6
    for n in range(gamma):
        p = _probs(gamma + 1, V, 300 + n, sparse=True)
        qd = _probs(1, V, 400 + n, sparse=True)[0]
        tokens = torch.full((gamma + 1,), 100, dtype=torch.int64)
        d = int(torch.multinomial(qd, 1, generator=gen))
        tokens[n + 1] = d
        if n % 2 == 0:
            p[n, d] = p[n].max()
            p[n] /= p[n].sum()
        u = torch.rand(2, generator=gen)
        acc, b = R.middle_accept(p, qd, d, n, u.tolist())
        td = tokens.to(DEV)
        ops.middle_accept(p.to(DEV), qd.to(DEV), td, u.to(DEV), n, gamma, out)
        assert out[:3].tolist() == [acc, b, d]
        if n + 1 + acc <= gamma:
            assert int(td[n + 1 + acc]) == b


@pytest.mark.parametrize("V", [512, 1024, 32000])
@pytest.mark.parametrize("T,P", [(0.6, 0.9), (1.0, 1e-9), (0.3, 0.5), (1.0, 0.999), (2.0, 0.95)])
def test_topp_probs_matches_oracle(V, T, P):
    """Fused temperature/top-p/softmax kernel vs the oracle's sort-based norm_logits (stable order)."""
    ops = _ops()
    lg = rnd(9, V, seed=90 + V, dtype=torch.float32) * 2.5
    lg[1] = lg[1].half().float()                       # fp16-valued logits (what the lm_head produces): exact ties
    lg[2, :5] = lg[2].max()                            # ties at the very top
    lg[3] = 0.0                                        # everything tied
    lg[4, 7] = 30.0                                    # one dominant token
    want = R.norm_logits(lg.clone(), T, -1, P)
    got = ops.topp_probs(lg.to(DEV), T, P).cpu()
    assert torch.isfinite(got).all() and (got.sum(-1) - 1).abs().max() < 1e-5
    for r in range(lg.shape[0]):
        sg, sw = got[r] > 0, want[r] > 0
        if torch.equal(sg, sw):
            Hh.close(got[r], want[r], rtol=2e-6, atol=1e-9)                        # measured: 6e-8 on p ~ 0.5 (an fp32 ulp)
        else:
            # the kept set may differ only at the top-p boundary, where the cumulative mass (summed in a
            # different fp32 order) is within rounding of top_p; ties must still resolve to the lower token id
            diff = torch.nonzero(sg != sw).flatten()
            assert diff.numel() <= 2, f"row {r}: kept sets differ in {diff.numel()} entries"
            p_full = torch.softmax(lg[r] / T, -1)
            order = torch.sort(p_full, descending=True, stable=True)
            cum = torch.cumsum(order.values.double(), 0)
            ranks = {int(t): i for i, t in enumerate(order.indices.tolist())}
            for t in diff.tolist():
                rk = ranks[t]
                before = float(cum[rk - 1]) if rk > 0 else 0.0
                assert abs(before - P) < 2e-5, f"row {r}: token {t} (rank {rk}) flipped far from the boundary ({before} vs {P})"
    if P < 1e-6:                                        # greedy emulation: one-hot on the LOWEST index among the maxima
        assert torch.equal(got, want)


def _topp_rows(rows, V, kind, seed):
    lg = rnd(rows, V, seed=seed, dtype=torch.float32) * 2.5
    if kind == "fp16":                                 # what an lm_head produces: fp16-valued logits, exact ties everywhere
        lg = lg.half().float()
    elif kind == "peaked":
        lg = (lg * 4).half().float()
    elif kind == "ties":
        lg = lg[:, :16].repeat(1, V // 16 + 1)[:, :V].contiguous()
    elif kind == "flat":
        lg = (lg * 0.01).half().float()
    elif kind == "equal":
        lg = torch.zeros_like(lg)
    elif kind == "ties20":                             # 20 equal dominant logits: top_p cuts the group inside a boundary bin of <= 64 entries
        lg = lg.half().float()
        lg[:, 100:120] = lg.max(-1, keepdim=True).values + 6.0
    return lg


@pytest.mark.parametrize("rows,V", [(1, 32000), (7, 32000), (8, 32768), (18, 32000), (3, 1024), (32, 4096)])
@pytest.mark.parametrize("kind", ["fp16", "peaked", "ties", "ties20", "flat", "equal"])
def test_topp_multi_workgroup_form_is_bit_identical_to_the_one_workgroup_kernel(rows, V, kind, monkeypatch):
    """tf_topp_probs_multi (every row over 16 / 8 workgroups of one launch, two in-launch hand-offs, candidates compacted per
    slice) against tf_topp_probs (one workgroup per row, itself pinned to the oracle above): torch.equal for fp16-valued rows,
    sharp rows, rows of 16 distinct values (2 000-entry tie groups cut at the boundary: radix tie ranking), 20 equal dominant logits (a cut tie
    group ranked by the in-wave finish), near-flat rows (every entry a candidate:
    the rounds walk global memory) and all-equal rows, at four (T, top_p) settings incl. top_p = 1 and the greedy emulation; with and
    without the per-panel maxima an lm_head epilogue hands over; twice in a row on the same control block."""
    ops = _ops()
    lg = _topp_rows(rows, V, kind, 300 + rows).to(DEV)
    pm = None
    if V % 16 == 0:
        pm = torch.full((V // 16, 32), float("-inf"), dtype=torch.float32, device=DEV)
        pm[:, :rows] = lg.view(rows, V // 16, 16).max(-1).values.t()
    from triforce_amd import hip
    L = hip.lib()
    st = ops._topp_multi(torch.device(DEV))

    def multi(T, P, panel):                            # the entry point itself (ops.topp_probs routes > 16 rows to the one-workgroup kernel)
        out = torch.empty_like(lg)
        hip.check(L.tf_topp_probs_multi(ops._ptr(lg), ops._ptr(panel), ops._ptr(out), rows, V, T, P, ops._ptr(st[0]), ops._ptr(st[1]),
                                        st[1].numel(), ops._stream()), "tf_topp_probs_multi")
        return out
    for T, P in ((0.6, 0.9), (1.0, 0.95), (0.8, 1.0), (1.0, 1e-9)):
        monkeypatch.setattr(ops, "TOPP_MULTI", False)
        want = ops.topp_probs(lg, T, P)
        monkeypatch.setattr(ops, "TOPP_MULTI", True)
        for panel, rowmax in ((None, 1), (None, 0), (pm, 1)):          # row maximum: from the whole row / through edge A / handed over
            L.tf_topp_multi_tune(3, rowmax)
            try:
                for _ in range(2):
                    got = multi(T, P, panel) if rows > 16 else ops.topp_probs(lg, T, P, panel_max=panel)
                    assert torch.equal(got, want), (kind, T, P, panel is not None, rowmax, int((got != want).sum()),
                                                    float((got - want).abs().max()))
            finally:
                L.tf_topp_multi_tune(3, 1)
    assert _ops_lib().tf_topp_multi_error(ops._ptr(st[0])) == 0


def _ops_lib():
    from triforce_amd import hip
    return hip.lib()


def test_topp_multi_lost_arrival_times_out_poisons_and_recovers(monkeypatch):
    ops, L = _ops(), _ops_lib()
    lg = _topp_rows(7, 32000, "fp16", 5).to(DEV)
    monkeypatch.setattr(ops, "TOPP_MULTI", True)
    want = ops.topp_probs(lg, 0.6, 0.9)
    st = ops._topp_multi(torch.device(DEV))
    old = L.tf_topp_multi_tune(0, 50)
    try:
        for edge in (0, 1):
            L.tf_topp_multi_tune(3, 0 if edge == 0 else 1)           # (edge A only exists when the slices exchange their maxima)
            L.tf_topp_multi_tune(1, edge + 1)
            bad = ops.topp_probs(lg, 0.6, 0.9)
            torch.cuda.synchronize()
            L.tf_topp_multi_tune(1, 0)
            assert L.tf_topp_multi_error(ops._ptr(st[0])) == edge + 1
            assert bool(torch.isnan(bad[0]).all())                   # the row whose arrival was lost; sticky from here on
            again = ops.topp_probs(lg, 0.6, 0.9)
            torch.cuda.synchronize()
            assert bool(torch.isnan(again).all())
            assert L.tf_topp_multi_reset(ops._ptr(st[0])) == 0
            good = ops.topp_probs(lg, 0.6, 0.9)
            assert torch.equal(good, want)
    finally:
        L.tf_topp_multi_tune(1, 0)
        L.tf_topp_multi_tune(3, 1)
        L.tf_topp_multi_tune(0, old)


def test_norm_logits_routes_to_fused_kernel_and_draft_row_shortcut():
    from triforce_amd.utils.sampling import norm_logits
    lg = rnd(7, 32000, seed=5, dtype=torch.float32).to(DEV) * 3
    full = norm_logits(lg, 0.6, -1, 0.9)
    last = norm_logits(lg[-1:], 0.6, -1, 0.9)[0]
    assert torch.equal(full[-1], last)                  # rows are independent
    want = R.norm_logits(lg.cpu(), 0.6, -1, 0.9)
    assert ((full.cpu() > 0) == (want > 0)).float().mean() > 0.9999


@pytest.mark.parametrize("M,N,K", [(7, 4096, 1376), (7, 4096, 2752), (7, 4096, 512), (1, 4096, 1376), (8, 4096, 5504),
                                   (7, 1536, 4096), (7, 5120, 1728), (7, 5120, 640)])
def test_skinny_gemm_at_tensor_parallel_shard_shapes(M, N, K):
    """Per-rank shapes of the 7B / 13B models at TP = 2, 4, 8 (TP_layers.py:126-147): down-proj K = I / W (1376 = 43
    tiles of 32 — odd and prime, so the K-split waves get unequal shares), o-proj K = H_local * D, q|k|v N = 3 * H_local
    * D.  Relative to an fp32 matmul of the same fp16 operands."""
    ops = _ops()
    x, w = rnd(M, K, seed=130 + M).to(DEV), rnd(N, K, seed=131, scale=0.05).to(DEV)
    got = ops.linear(x, ops.PackedLinear(w)).float()
    want = x.float() @ w.float().t()
    assert ((got - want).abs() / (want.abs() + 1)).max().item() < 5e-3


@pytest.mark.parametrize("M,I,K", [(7, 1376, 4096), (7, 2752, 4096), (1, 1376, 4096), (7, 5504, 4096)])
def test_skinny_swiglu_at_tensor_parallel_shard_shapes(M, I, K):
    ops = _ops()
    x, wgu = rnd(M, K, seed=140 + M).to(DEV), rnd(2 * I, K, seed=141, scale=0.05).to(DEV)
    got = ops.mlp_act(x, ops.PackedLinear(wgu, split=2)).float()
    g = (x.float() @ wgu[:I].float().t()).half().float()
    u = (x.float() @ wgu[I:].float().t()).half().float()
    want = torch.nn.functional.silu(g).half().float() * u
    assert ((got - want).abs() / (want.abs() + 1)).max().item() < 1e-2


@pytest.mark.parametrize("M", [1, 7, 8, 16, 17, 32])
@pytest.mark.parametrize("N,K", [(768, 768), (2304, 768), (4096, 4096), (12288, 4096), (4096, 11008), (32000, 768)])
def test_skinny_gemm_matches_linear(M, N, K):
    """Hand-written weight-streaming GEMM (pre-packed weights) vs the oracle's F.linear."""
    ops = _ops()
    x, w = rnd(M, K, seed=110 + M), rnd(N, K, seed=111, scale=0.05)
    want = R.linear(x, w)
    pl = ops.PackedLinear(w.to(DEV))
    assert pl.wp is not None
    got = ops.linear(x.to(DEV), pl)
    # fp32 accumulation in a different order than the CPU GEMM: <=1 fp16 ulp on a small fraction of outputs
    ulp_report("skinny_gemm", got, want, max_ulp_frac=3e-2, atol=1e-4)
    got32 = ops.linear(x.to(DEV), pl, out_f32=True)
    assert got32.dtype == torch.float32 and torch.equal(got32.cpu(), got.float().cpu())   # fp16 GEMM, then the cast
    # pack/unpack is a pure permutation
    back = pl.wp.view(N // 16, K // 32, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(N, K)
    assert torch.equal(back.cpu(), w)


@pytest.mark.parametrize("M,I,K", [(1, 3072, 768), (7, 11008, 4096), (18, 1728, 5120), (32, 768, 256)])
def test_skinny_gemm_swiglu_matches_oracle(M, I, K):
    ops = _ops()
    x, wgu = rnd(M, K, seed=120 + M), rnd(2 * I, K, seed=121, scale=0.05)
    gu = R.linear(x, wgu)
    want = R.silu_mul(gu[:, :I], gu[:, I:])
    pl = ops.PackedLinear(wgu.to(DEV), split=2)
    got = ops.mlp_act(x.to(DEV), pl)
    # (1) the fused epilogue, given the SAME fp16 gate/up the un-fused kernels produce (identical panel / chunk /
    #     wave split => identical accumulation), must equal the oracle's silu*up up to expf's last bit
    gate16 = ops.linear(x.to(DEV), ops.PackedLinear(wgu[:I].to(DEV)))
    up16 = ops.linear(x.to(DEV), ops.PackedLinear(wgu[I:].to(DEV)))
    #     (few-panel shapes run the un-fused GEMM with more K-splits per panel than the fused one — SG_WAVES_WIDE,
    #     csrc/gemv.hip — so their gate/up may differ by an fp16 rounding; the strict check covers the other shapes)
    same_split = (I // 16) > 512 or (K // 32) < 16
    if same_split:
        ulp_report("swiglu epilogue", got, R.silu_mul(gate16.cpu(), up16.cpu()), max_ulp_frac=2e-2, ulps=1)
    # (2) against the pure CPU pipeline: gate/up may each land on the neighbouring fp16 (different fp32 summation
    #     order) and silu amplifies a relative gate error by |1 + g(1-sigmoid(g))| (up to ~4x for g ~ -4), so the
    #     bound is absolute in terms of the inputs' spacing rather than a few ulp of the product
    d = (got.float().cpu() - want.float()).abs()
    bound = 2.0 ** -9 * (1.0 + gu[:, :I].float().abs()) * (1.0 + gu[:, I:].float().abs())
    assert (d <= bound).all(), f"swiglu vs CPU pipeline: max excess {(d - bound).max():.3e}"
    assert (d > 0).float().mean() < 6e-2
    big = ops.mlp_act(rnd(40, K, seed=5).to(DEV), pl)                 # >32 rows: hipBLASLt + silu_mul path
    assert big.shape == (40, I)


# ------------------------------------------------------------------------------------------
# fused forms of the skinny GEMM (csrc/gemv.hip): norm prologue, residual / RoPE+append epilogues
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,H,D,K", [(1, 2, 128, 256), (7, 32, 128, 4096), (18, 4, 64, 768), (32, 3, 128, 512),
                                     (9, 12, 64, 768)])
@pytest.mark.parametrize("rotate_k", [True, False])
def test_qkv_rope_fused(M, H, D, K, rotate_k):
    ops = _ops()
    eps, slot0, T = 1e-5, 5, 64
    w = rnd(3 * H * D, K, seed=200 + M, scale=0.05)
    x = rnd(M, K, seed=201)
    ln = (1 + 0.1 * rnd(K, seed=202).float()).half()
    cos, sin = R.rope_tables_yarn(D, 4096, 16.0, 256) if D == 128 else R.rope_tables_plain(D, 4096)
    pos = torch.randint(0, 4096, (M,), generator=torch.Generator().manual_seed(M))
    pl = ops.PackedLinear(w.to(DEV), rope=(H, D))
    assert pl.wp_rope is not None
    cd, sd, pd = cos.to(DEV), sin.to(DEV), pos.to(DEV)

    def caches():
        return (torch.zeros(H, T, D, dtype=torch.float16, device=DEV), torch.zeros(H, T, D, dtype=torch.float16, device=DEV))
    # (1) RoPE + append epilogue alone (input already normalised): bit-exact vs GEMM kernel + tf_rope_append —
    #     the packed row order changes which panel a row sits in, not how its dot product is accumulated
    h = ops.rmsnorm(x.to(DEV), ln.to(DEV), eps)
    k1, v1 = caches()
    k2, v2 = caches()
    q1 = ops.qkv_rope(h, pl, None, 0.0, cd, sd, pd, k1, v1, slot0, H, D, rotate_k=rotate_k)
    q2 = ops.rope_append(ops.linear(h, pl), cd, sd, pd, k2, v2, slot0, H, D, rotate_k=rotate_k)
    assert torch.equal(q1, q2) and torch.equal(k1, k2) and torch.equal(v1, v2)
    # (2) with the RMSNorm prologue, against the oracle pipeline (norm -> linear -> rope)
    k3, v3 = caches()
    sdev = torch.tensor([slot0], dtype=torch.int32, device=DEV)
    q3 = ops.qkv_rope(x.to(DEV), pl, ln.to(DEV), eps, cd, sd, pd, k3, v3, 0, H, D, rotate_k=rotate_k, slot0_dev=sdev)
    qkv = R.linear(R.rms_norm(x, ln, eps), w)
    wq = R.apply_rope(qkv[:, :H * D].view(M, H, D), cos, sin, pos)
    wk = qkv[:, H * D:2 * H * D].view(M, H, D)
    wk = R.apply_rope(wk, cos, sin, pos) if rotate_k else wk
    wv = qkv[:, 2 * H * D:].view(M, H, D)
    # fp32 sums in a different order (norm and GEMM) -> neighbouring fp16 on a few elements, then two more fp16
    # roundings in the rotation: <= 2 ulp (+1e-3 abs near zero crossings of x*cos + rot*sin)
    ulp_report("qkv_rope q", q3, wq, max_ulp_frac=8e-2, ulps=2, atol=1e-3)
    ulp_report("qkv_rope k", k3[:, slot0:slot0 + M].permute(1, 0, 2), wk, max_ulp_frac=8e-2, ulps=2, atol=1e-3)
    ulp_report("qkv_rope v", v3[:, slot0:slot0 + M].permute(1, 0, 2), wv, max_ulp_frac=5e-2, ulps=1, atol=1e-4)
    assert k3[:, :slot0].abs().sum() == 0 and k3[:, slot0 + M:].abs().sum() == 0


@pytest.mark.parametrize("M,N,K", [(1, 4096, 4096), (8, 4096, 11008), (18, 768, 3072), (32, 512, 256), (7, 32000, 4096)])
def test_skinny_gemm_norm_prologue_and_residual_epilogue(M, N, K):
    ops = _ops()
    eps = 1e-5
    w = rnd(N, K, seed=210 + M, scale=0.05)
    x, res = rnd(M, K, seed=211), rnd(M, N, seed=212)
    ln = (1 + 0.1 * rnd(K, seed=213).float()).half()
    pl = ops.PackedLinear(w.to(DEV))
    xd, resd, lnd = x.to(DEV), res.to(DEV), ln.to(DEV)
    # residual epilogue: bit-exact vs the plain kernel followed by an fp16 add; in place on the residual buffer
    y = ops.linear(xd, pl)
    buf = resd.clone()
    out = ops.linear(xd, pl, resid=buf, out=buf)
    assert out.data_ptr() == buf.data_ptr() and torch.equal(buf, resd + y)
    # norm prologue vs the stand-alone RMSNorm kernel + GEMM (same device, different fp32 summation order of the
    # sum of squares) and vs the oracle
    yn = ops.linear(xd, pl, ln=lnd, eps=eps)
    ulp_report("norm prologue vs kernels", yn, ops.linear(ops.rmsnorm(xd, lnd, eps), pl), max_ulp_frac=3e-2, ulps=1, atol=1e-4)
    # (a different fp16 neighbour of a normalised input moves a dot product by up to ~1 ulp on top of the GEMM's own)
    #  — and the absolute error of a K-term fp16 dot product does not shrink with a small result (cancellation):
    #  atol = one fp16 spacing at the typical |y| ~ 2..4
    ulp_report("norm prologue vs oracle", yn, R.linear(R.rms_norm(x, ln, eps), w), max_ulp_frac=5e-2, ulps=2, atol=2e-3)
    if N <= 4096:
        y32 = ops.linear(xd, pl, out_f32=True, ln=lnd, eps=eps)
        assert torch.equal(y32, yn.float())
    # both at once, plus the sum-of-squares hand-off: the residual GEMM leaves per-panel sum(y^2), a following norm
    # prologue that folds those partials must agree with one that re-reads its input
    buf = resd.clone()
    ss = ops.ss_buffer(N, DEV)
    ops.linear(xd, pl, ln=lnd, eps=eps, resid=buf, out=buf, ss_out=ss)
    assert torch.equal(buf, resd + yn)
    rows_ss = ss[:, :M].sum(dim=0)
    Hh.close(rows_ss, (buf.float() ** 2).sum(dim=1), rtol=2e-6, atol=1e-3)    # measured: 1.6e-7 relative (fp32 sums of 4 096 squares)
    if N % 32 == 0 and N <= 11008:
        w2 = rnd(256, N, seed=214, scale=0.05)
        ln2 = (1 + 0.1 * rnd(N, seed=215).float()).half().to(DEV)
        pl2 = ops.PackedLinear(w2.to(DEV))
        ya = ops.linear(buf, pl2, ln=ln2, eps=eps, ss_in=ss)
        yb = ops.linear(buf, pl2, ln=ln2, eps=eps)
        ulp_report("ss hand-off vs two-pass prologue", ya, yb, max_ulp_frac=3e-2, ulps=2, atol=1e-4)


@pytest.mark.parametrize("M,I,K", [(1, 3072, 768), (7, 11008, 4096), (18, 1728, 5120)])
def test_swiglu_norm_prologue(M, I, K):
    ops = _ops()
    eps = 1e-5
    x, wgu = rnd(M, K, seed=220 + M), rnd(2 * I, K, seed=221, scale=0.05)
    ln = (1 + 0.1 * rnd(K, seed=222).float()).half()
    pl = ops.PackedLinear(wgu.to(DEV), split=2)
    got = ops.mlp_act(x.to(DEV), pl, ln=ln.to(DEV), eps=eps)
    ref = ops.mlp_act(ops.rmsnorm(x.to(DEV), ln.to(DEV), eps), pl)
    # vs the stand-alone RMSNorm kernel + fused SwiGLU GEMM: a normalised input landing on the neighbouring fp16 moves
    # gate / up by ~1 ulp, silu amplifies the gate's share -> a few ulp of the product on a few % of the elements
    d = (got.float() - ref.float()).abs()
    tol = 4 * ref.float().abs() * 2 ** -10 + 2e-3
    assert (d > 0).float().mean() < 8e-2 and bool((d <= tol).all()), (float((d > 0).float().mean()), float(d.max()))
    h = R.rms_norm(x, ln, eps)
    gu = R.linear(h, wgu)
    want = R.silu_mul(gu[:, :I], gu[:, I:])
    dd = (got.float().cpu() - want.float()).abs()
    tolw = 6 * want.float().abs() * 2 ** -10 + 4e-3
    assert bool((dd <= tolw).all()) and dd.mean() < 5e-4, (float(dd.max()), float(dd.mean()))


@pytest.mark.parametrize("M", [1, 7, 17, 32])
@pytest.mark.parametrize("kind,N,K,HD", [("qkv", 1536, 4096, (4, 128)), ("qkv", 1920, 5120, (5, 128)),
                                          ("swiglu", 1376, 4096, None), ("swiglu", 1728, 5120, None),
                                          ("plain", 3072, 4096, None), ("plain", 1024, 11008, None)])
def test_gemm_split_across_workgroups(M, kind, N, K, HD, monkeypatch):
    """Few-panel GEMMs (the q|k|v / gate|up shards of a tensor-parallel rank: models/TP_layers.py:126-147) CAN split K
    across up to 4 workgroups per panel, the partial sums meeting through the registered workspace (csrc/gemv.hip
    SgKsplit) — built and measured in round 4, no gain in situ, so the launch rule leaves it off (tf_sg_tune key 3 = 0);
    this test keeps the form correct for the next chip / shape where it may pay.
    Against the one-workgroup form: same fp16 results up to the fp32 re-association of the K sum (<= 1 ulp on a few %);
    run to run: bit-identical (the last arriver adds the partials in split order, whoever it is); tickets back to
    zero; norm prologue both ways (folding the hand-off / re-reading x over ALL of K); inside a hipGraph."""
    ops = _ops()
    from triforce_amd import hip
    L = hip.lib()
    eps = 1e-5
    monkeypatch.setattr(ops, "N8_ENABLED", False)             # (these shapes take the narrow-panel form since round 5)
    x = rnd(M, K, seed=400 + M).to(DEV)
    ln = (1 + 0.1 * rnd(K, seed=401).float()).half().to(DEV)
    ssx = ops.ss_buffer(K, DEV)
    ssx[:, :M] = x.float().square().view(M, K // 16, 16).sum(-1).t()
    if kind == "qkv":
        H, D = HD
        w = rnd(N, K, seed=402, scale=0.05).to(DEV)
        pl = ops.PackedLinear(w, rope=(H, D))
        cos, sin = (t.to(DEV) for t in R.rope_tables_yarn(D, 4096, 16.0, 256))
        pos = torch.randint(0, 4096, (M,), generator=torch.Generator().manual_seed(M)).to(DEV)

        def run(ss_in=None):
            k = torch.zeros(H, 64, D, dtype=torch.float16, device=DEV)
            v = torch.zeros(H, 64, D, dtype=torch.float16, device=DEV)
            q = ops.qkv_rope(x, pl, ln, eps, cos, sin, pos, k, v, 3, H, D, ss_in=ss_in)
            return torch.cat([q.reshape(-1), k.reshape(-1), v.reshape(-1)])
    elif kind == "swiglu":
        pl = ops.PackedLinear(rnd(2 * N, K, seed=403, scale=0.05).to(DEV), split=2)

        def run(ss_in=None):
            return ops.mlp_act(x, pl, ln=ln, eps=eps, ss_in=ss_in).reshape(-1)
    else:
        pl = ops.PackedLinear(rnd(N, K, seed=404, scale=0.05).to(DEV))
        res = rnd(M, N, seed=405).to(DEV)

        def run(ss_in=None):
            buf, ss = res.clone(), ops.ss_buffer(N, DEV)
            ops.linear(x, pl, ln=ln, eps=eps, ss_in=ss_in, resid=buf, out=buf, ss_out=ss)
            return torch.cat([buf.reshape(-1), ss[:, :M].reshape(-1).half()])
    assert torch.device(DEV) in ops._SG_WS, "split-K workspace was not registered"
    one = [run(), run(ssx)]                                   # the shipped rule at these shapes: one workgroup per panel
    old = L.tf_sg_tune(3, 200)                                # split below 200 panel groups
    try:
        split = [run(), run(ssx)]
        again = [run(), run(ssx)]
        torch.cuda.synchronize()
        assert torch.equal(split[0], again[0]) and torch.equal(split[1], again[1])
        ws = ops._SG_WS[torch.device(DEV)]                    # 4 per-stream slots, each headed by 16 KiB of tickets
        assert all(int(ws[i * (ws.numel() // 4):i * (ws.numel() // 4) + 16384].sum()) == 0 for i in range(4)), "tickets were not left zero"
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            cap = run()
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(cap, split[0])
    finally:
        L.tf_sg_tune(3, old)
    for a, b in zip(split, one):
        d = (a.float() - b.float()).abs()
        # a re-associated K sum moves a pre-RoPE value by at most one fp16 ulp; x * cos + rotate_half(x) * sin then adds two
        # such values of magnitude up to ~8 with opposite signs, so the rotated q / k rows get an ABSOLUTE slack of two
        # spacings at that magnitude (1.6e-2) where the other forms get 2 ulp of the result
        tol = 2 * b.float().abs() * 2 ** -10 + (1.6e-2 if kind == "qkv" else 2e-3)
        assert bool((d <= tol).all()) and float((d > 0).float().mean()) < 0.12, (float(d.max()), float((d > 0).float().mean()))


@pytest.mark.parametrize("M", [1, 7, 8, 9, 16, 17, 24, 25])
@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("kind,N,K,HD", [("qkv", 1536, 4096, (4, 128)), ("qkv", 1920, 5120, (5, 128)), ("qkv", 768, 1024, (4, 64)),
                                          ("swiglu", 1376, 4096, None), ("swiglu", 1728, 5120, None)])
def test_narrow_panel_norm_gemms(M, packed, kind, N, K, HD, monkeypatch):
    """Round 5: the q|k|v / gate|up shards of a tensor-parallel rank (models/TP_layers.py:126-147: 3 * H/W * D and I/W rows —
    96 / 86 16-row panels at 7B with 8 ranks, 120 / 108 at 13B) run 8-ROW panels (csrc/gemv.hip skinny_gemm_n8_kernel: two
    k-chunks stacked on the MFMA's 16 A rows, the diagonal 8 x 8 blocks summed) so that twice the workgroups stream half the
    bytes each.  Checked (a) against the oracle pipeline norm -> linear -> RoPE / SwiGLU at the tolerances of the 16-row
    tests (test_qkv_rope_fused, test_swiglu_norm_prologue), (b) against the 16-row kernel on the same device — same
    rounding points, K summed in another order: <= 1 fp16 ulp before the epilogue —, (c) sum-of-squares hand-off == re-reading
    x, (d) run to run and inside a hipGraph bit-identical, (e) 25 rows fall back to the 16-row form.  Row-major and
    k-octet-major activations."""
    ops = _ops()
    eps = 1e-5
    x = rnd(M, K, seed=500 + M)
    ln = (1 + 0.1 * rnd(K, seed=501).float()).half()
    xd, lnd = x.to(DEV), ln.to(DEV)
    xin = ops.Act.from_rows(xd) if packed else xd
    ssx = ops.ss_buffer(K, DEV)
    ssx[:, :M] = xd.float().square().view(M, K // 16, 16).sum(-1).t()
    if kind == "qkv":
        H, D = HD
        w = rnd(N, K, seed=502, scale=0.05)
        cos, sin = R.rope_tables_yarn(D, 4096, 16.0, 256) if D == 128 else R.rope_tables_plain(D, 4096)
        pos = torch.randint(0, 4096, (M,), generator=torch.Generator().manual_seed(M))
        cd, sd, pd = cos.to(DEV), sin.to(DEV), pos.to(DEV)

        def build():
            return ops.PackedLinear(w.to(DEV), rope=(H, D))

        def run(pl, ss_in=None):
            k = torch.zeros(H, 64, D, dtype=torch.float16, device=DEV)
            v = torch.zeros(H, 64, D, dtype=torch.float16, device=DEV)
            q = ops.qkv_rope(xin, pl, lnd, eps, cd, sd, pd, k, v, 3, H, D, ss_in=ss_in)
            if not torch.cuda.is_current_stream_capturing():
                assert k[:, :3].abs().sum() == 0 and k[:, 3 + M:].abs().sum() == 0
            return q, k[:, 3:3 + M].permute(1, 0, 2).contiguous(), v[:, 3:3 + M].permute(1, 0, 2).contiguous()
    else:
        wgu = rnd(2 * N, K, seed=503, scale=0.05)

        def build():
            return ops.PackedLinear(wgu.to(DEV), split=2)

        def run(pl, ss_in=None):
            a = ops.mlp_act(xin, pl, ln=lnd, eps=eps, ss_in=ss_in)
            return (a.rows() if packed else a,)
    pl8 = build()
    assert (pl8.wp_rope_n8 if kind == "qkv" else pl8.parts_n8) is not None, "narrow-panel copy was not packed"
    monkeypatch.setattr(ops, "N8_ENABLED", False)
    pl16 = build()
    assert pl16.wp_rope_n8 is None and pl16.parts_n8 is None
    got, got_ss, wide = run(pl8), run(pl8, ssx), run(pl16)
    again = run(pl8)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        cap = run(pl8, ssx)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    for a, b, c, d in zip(got, again, cap, got_ss):
        assert torch.equal(a, b), "run to run"
        assert torch.equal(c, d), "captured == eager"
    if M > ops.N8_MAX_ROWS:                                    # (e) 16-row form: bit-identical to the copy without n8
        for a, b in zip(got, wide):
            assert torch.equal(a, b)
    h = R.rms_norm(x, ln, eps)
    if kind == "qkv":
        qkv = R.linear(h, w)
        wq = R.apply_rope(qkv[:, :H * D].view(M, H, D), cos, sin, pos)
        wk = R.apply_rope(qkv[:, H * D:2 * H * D].view(M, H, D), cos, sin, pos)
        wv = qkv[:, 2 * H * D:].view(M, H, D)
        # rotated q / k = a cos + b sin of two pre-RoPE values that may EACH sit on the neighbouring fp16 (another fp32 order of
        # the K sum than the CPU's): where the two terms cancel, the error is absolute — two spacings at the largest pre-RoPE
        # magnitude of the block
        spacing = 2.0 ** (math.floor(math.log2(float(qkv[:, :2 * H * D].float().abs().max()))) - 10)
        for res in (got, got_ss):
            ulp_report("n8 qkv_rope q", res[0], wq, max_ulp_frac=8e-2, ulps=2, atol=2 * spacing)
            ulp_report("n8 qkv_rope k", res[1], wk, max_ulp_frac=8e-2, ulps=2, atol=2 * spacing)
            # (v = the bare norm-prologue GEMM: the bound of test_skinny_gemm_norm_prologue_and_residual_epilogue — the 16-row
            #  kernel, which serves the 25-row case here, needs it at K = 5120 just the same)
            ulp_report("n8 qkv_rope v", res[2], wv, max_ulp_frac=5e-2, ulps=2, atol=2e-3)
        # vs the 16-row kernel: v is the bare GEMM (<= 1 ulp), rotated q / k add two such values (see the K-split test)
        # (two fp16 roundings of sums that differ in fp32 order: one ulp — of the LARGER binade when they straddle a power of two)
        #  + an ABSOLUTE part: the two kernels sum x^2 in different orders, a few normalised inputs land on the neighbouring
        #  fp16, and each moves the dot product by ~|w| * ulp(h) ~ 5e-5 whatever the size of the result)
        ulp_report("n8 vs 16-row v", got[2], wide[2], max_ulp_frac=0.25, ulps=2, atol=2e-3)
        for a, b in zip(got[:2], wide[:2]):
            dq = (a.float() - b.float()).abs()
            assert bool((dq <= 2 * b.float().abs() * 2 ** -10 + 2 * spacing).all()) and float((dq > 0).float().mean()) < 0.25
    else:
        gu = R.linear(h, wgu)
        want = R.silu_mul(gu[:, :N], gu[:, N:])
        for res in (got, got_ss):
            dd = (res[0].float().cpu() - want.float()).abs()
            tolw = 6 * want.float().abs() * 2 ** -10 + 4e-3
            assert bool((dd <= tolw).all()) and dd.mean() < 5e-4, (float(dd.max()), float(dd.mean()))
        dq = (got[0].float() - wide[0].float()).abs()
        assert bool((dq <= 6 * wide[0].float().abs() * 2 ** -10 + 4e-3).all()) and float((dq > 0).float().mean()) < 0.15
        spacing = 2e-3
    # (c) folding the producer's partials vs re-reading x: the sum of squares in another fp32 order (a few normalised inputs on
    #     the neighbouring fp16; silu amplifies the gate's share: the oracle bound of the SwiGLU form)
    for a, b in zip(got, got_ss):
        dq = (a.float() - b.float()).abs()
        assert bool((dq <= (2 if kind == "qkv" else 6) * b.float().abs() * 2 ** -10 + 2 * spacing).all())


def test_cursor_forms_of_the_sampling_kernels_equal_the_pointer_forms():
    """Round 5: inside the inner-iteration hipGraphs the draw / accept kernels read their uniforms as ubuf[cursor + k]
    (tf_*_cur; utils/decoding.py:163-223 as one launch per iteration) and advance the device cursor themselves.  Same
    kernels, same numbers: records, written tokens and sampled ids must equal the pointer forms bit for bit, and the cursor
    must move by exactly what each decision consumed."""
    ops = _ops()
    V, gamma = 32000, 6
    g = torch.Generator().manual_seed(77)
    ubuf = torch.rand(512, generator=g).to(DEV)
    for trial in range(6):
        p = torch.softmax(torch.randn(gamma + 2, V, generator=g) * 2, -1).to(DEV)
        q = torch.softmax(torch.randn(gamma + 1, V, generator=g) * 2, -1).to(DEV)
        if trial % 2:                                             # near-identical rows: accepted drafts, long chains
            q = (p[:gamma + 1] * 0.98 + 0.02 / V).contiguous()
        base = 5 + 17 * trial
        cur = torch.tensor([base], dtype=torch.int64, device=DEV)
        # draw
        t_ptr = torch.zeros(1, dtype=torch.int64, device=DEV)
        t_cur = torch.zeros(1, dtype=torch.int64, device=DEV)
        ops.sample_inverse_cdf(q[0], ubuf[base:base + 1], t_ptr)
        ops.sample_inverse_cdf_cur(q[0], ubuf, cur, 0, t_cur)
        assert torch.equal(t_ptr, t_cur) and int(cur) == base
        # inner accept at every position n
        for n in range(gamma):
            toks = torch.randint(0, V, (gamma + 1,), generator=g).to(DEV)
            toks[n + 1] = int(torch.multinomial(q[n].cpu(), 1))
            ta, tb = toks.clone(), toks.clone()
            ra = torch.full((4,), -1, dtype=torch.int64, device=DEV)
            rb = torch.full((4,), -1, dtype=torch.int64, device=DEV)
            cur.fill_(base)
            ops.middle_accept(p, q[n], ta, ubuf[base + 1:base + 3], n, gamma, ra)
            ops.middle_accept_cur(p, q[n], tb, ubuf, cur, n, gamma, rb)
            assert torch.equal(ra[:3], rb[:3]) and torch.equal(ta, tb), (trial, n)
            assert int(rb[3]) == base and int(cur) == base + 3
        # outer chain
        for g2 in (gamma, gamma + 1):
            toks = torch.stack([torch.multinomial(q[i].cpu(), 1)[0] for i in range(g2)]).to(DEV)
            ra = torch.zeros(4, dtype=torch.int64, device=DEV)
            rb = torch.zeros(4, dtype=torch.int64, device=DEV)
            cur.fill_(base)
            ops.accept_chain(p, q, toks, ubuf[base:base + g2 + 1], g2, False, 2, ra)
            ops.accept_chain_cur(p, q, toks, ubuf, cur, g2, False, 2, rb)
            assert torch.equal(ra, rb), (trial, g2, ra.tolist(), rb.tolist())
            assert int(cur) == base + int(rb[3])
            # ... and the form that also prepares what follows its record (tf_accept_chain_step): same record, the pass tokens
            # of reference decoding.py:137 in the token buffer, the next verify's positions / slot / key count for both lengths
            for eos in (2, int(toks[min(1, g2 - 1)])):            # an eos inside the chain: pad instead of a correction token
                S = 1000 + 13 * trial
                tok_buf = torch.cat([torch.tensor([777], device=DEV), toks, torch.full((gamma + 3 - 1 - g2,), -5, device=DEV)])
                want_rec = torch.zeros(4, dtype=torch.int64, device=DEV)
                cur.fill_(base)
                ops.accept_chain_cur(p, q, toks, ubuf, cur, g2, False, eos, want_rec)
                count, nxt, reason, _ = want_rec.tolist()
                sets = [(torch.zeros(ql, dtype=torch.int64, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV),
                         torch.zeros(1, dtype=torch.int32, device=DEV), ql) for ql in (gamma + 1, gamma + 2)]
                s_src = torch.tensor([S], dtype=torch.int32, device=DEV)
                rc = torch.zeros(4, dtype=torch.int64, device=DEV)
                cur.fill_(base)
                ops.accept_chain_step(p, q, tok_buf, ubuf, cur, g2, False, eos, 100, s_src, sets, rc)
                assert torch.equal(rc, want_rec) and int(cur) == base + int(rc[3])
                want_tok = [777] + toks[:count].tolist() + ([nxt] if reason != 2 else [100]) + [100] * (g2 - count)
                assert tok_buf[:g2 + 2].tolist() == want_tok, (tok_buf.tolist(), want_tok, count, reason)
                assert tok_buf[g2 + 2:].tolist() == [-5] * (gamma + 3 - g2 - 2)
                for pos, slot, sk, ql in sets:
                    assert pos.tolist() == list(range(S + count + 1, S + count + 1 + ql))
                    assert int(slot) == S + count + 1 and int(sk) == S + count + 1 + ql


def test_mid_record_tokens_equals_the_torch_form():
    """The tensor-parallel loop applies RANK 0's broadcast middle_accept record to the token buffer (reference
    utils/decoding.py:452-470: the accepted drafted token stays, the follow-up goes behind it; rejected: the follow-up replaces
    it) — tf_mid_record_tokens in one launch against the torch statement it replaces, every position, both outcomes, the end of
    the buffer."""
    ops = _ops()
    from triforce_amd.utils.decoding import _mid_tokens
    gamma = 6
    for n in range(gamma):
        for acc in (0, 1):
            rec = torch.tensor([acc, 4242, 1717, 0], dtype=torch.int64, device=DEV)
            a = torch.arange(100, 100 + gamma + 1, dtype=torch.int64, device=DEV)
            b = a.clone()
            a[n + 1:n + 3].copy_(_mid_tokens(rec, n, gamma, a))
            ops.mid_record_tokens(rec, b, n)
            assert torch.equal(a, b), (n, acc, a.tolist(), b.tolist())


def test_launch_plans_equal_the_per_call_wrappers():
    """Round 5: the decode loop's per-step launches over fixed buffers (retrieval-tail copy, draft-window shift, token /
    position set-up) go through plans that validate their tensors once (ops.KvCopyPairPlan / KvShiftPairPlan / SetTokensPlan).
    Same launches, same bytes as the per-call wrappers, and the same refusals of row ranges that leave the caches."""
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    src_k = torch.randn(3, 2, 40, 64, generator=g).half().to(DEV)
    src_v = torch.randn(3, 2, 40, 64, generator=g).half().to(DEV)
    for args in ((30, 4, 9), (0, 0, 16), (39, 15, 1)):
        a_k, a_v = torch.zeros(3, 2, 16, 64, dtype=torch.float16, device=DEV), torch.zeros(3, 2, 16, 64, dtype=torch.float16, device=DEV)
        b_k, b_v = a_k.clone(), a_v.clone()
        ops.kv_copy_rows_pair(src_k, src_v, a_k, a_v, *args)
        ops.KvCopyPairPlan(src_k, src_v, b_k, b_v)(*args)
        assert torch.equal(a_k, b_k) and torch.equal(a_v, b_v) and a_k.abs().sum() > 0
    plan = ops.KvCopyPairPlan(src_k, src_v, a_k, a_v)
    for args in ((33, 0, 8), (0, 9, 8), (-1, 0, 2)):
        with pytest.raises(IndexError):
            plan(*args)
    plan(0, 0, 0)                                                # nothing to copy: no launch, no error
    for args in ((8, 0, 30), (10, 4, 30), (12, 0, 28)):           # (the window only ever moves DOWN: evict_for_spec)
        c_k, c_v = src_k.clone(), src_v.clone()
        d_k, d_v = src_k.clone(), src_v.clone()
        ops.kv_shift_rows_pair(c_k, c_v, *args)
        ops.KvShiftPairPlan(d_k, d_v)(*args)
        assert torch.equal(c_k, d_k) and torch.equal(c_v, d_v)
    with pytest.raises(IndexError):
        ops.KvShiftPairPlan(src_k, src_v)(20, 0, 30)
    dst = torch.zeros(9, dtype=torch.int64, device=DEV)
    pos = torch.zeros(7, dtype=torch.int64, device=DEV)
    slot, sk = torch.zeros(1, dtype=torch.int32, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV)
    d2, p2, s2, k2 = dst.clone(), pos.clone(), slot.clone(), sk.clone()
    ops.set_tokens(dst, [5, 6, 7], 100, pos=pos, pos0=1234, slot=slot, sk=sk, sk_val=1241)
    ops.SetTokensPlan(d2, p2, s2, k2)([5, 6, 7], 100, pos0=1234, sk_val=1241)
    assert torch.equal(dst, d2) and torch.equal(pos, p2) and torch.equal(slot, s2) and torch.equal(sk, k2)
    assert dst.tolist() == [5, 6, 7] + [100] * 6 and pos.tolist() == list(range(1234, 1241)) and int(slot) == 1234 and int(sk) == 1241
    d3 = torch.full((9,), -1, dtype=torch.int64, device=DEV)
    ops.SetTokensPlan(d3)([1, 2], 100, n_dst=5)                   # only the first n_dst entries are written
    assert d3.tolist() == [1, 2, 100, 100, 100, -1, -1, -1, -1]


def test_row_copy_wrappers_refuse_out_of_range_rows():
    """kv_copy_rows / kv_shift_rows / kv_gather_rows move whole token rows with no bounds check on the device: the
    wrappers refuse ranges that leave the tensors (a prompt longer than the declared prefill, a run past the
    generated-token capacity) instead of overwriting neighbouring device memory."""
    ops = _ops()
    src = torch.zeros(2, 3, 16, 64, dtype=torch.float16, device=DEV)
    dst = torch.zeros(2, 3, 8, 64, dtype=torch.float16, device=DEV)
    ops.kv_copy_rows(src, dst, 8, 0, 8)                          # exactly fits
    for args in ((8, 1, 8), (9, 0, 8), (0, 0, 9), (-1, 0, 2)):
        with pytest.raises(IndexError):
            ops.kv_copy_rows(src, dst, *args)
    ops.kv_shift_rows(src, 8, 0, 8)
    with pytest.raises(IndexError):
        ops.kv_shift_rows(src, 9, 0, 8)
    with pytest.raises(IndexError):
        ops.kv_shift_rows(src, 0, 12, 8)
    idx = torch.tensor([0, 2, 5], dtype=torch.int32, device=DEV)
    ops.kv_gather_rows(src, src.clone(), 10, idx, max_index=5)
    with pytest.raises(IndexError):
        ops.kv_gather_rows(src, src.clone(), 10, idx, max_index=6)   # source row 16 does not exist
    with pytest.raises(IndexError):
        ops.kv_gather_rows(src, src.clone(), 14, idx)                # destination rows 14..16


def test_attn_fused_merge_never_folds_stale_partials_under_uneven_load(monkeypatch):
    """The one-launch split merge hands partials from 8 workgroups (on 8 XCDs) to the last arriver through write-through
    stores, a drained store queue (s_waitcnt vmcnt(0)) and a ticket.  Round 2 shipped it without the drain — a narrow
    window in which the last arriver could fold the PREVIOUS launch's partials.  Stress: 600 back-to-back launches whose
    query changes every launch (stale partials would show), with a bandwidth hog running on a second stream so that the
    workgroups of a head finish at uneven times; every output must equal the two-launch form bit for bit."""
    ops = _ops()
    sq, sk, H, D = 7, 4103, 32, 128
    scale = R.softmax_scale_for(D)
    g = torch.Generator(device=DEV).manual_seed(77)
    k = torch.randn(H, sk, D, generator=g, device=DEV, dtype=torch.float16)
    v = torch.randn(H, sk, D, generator=g, device=DEV, dtype=torch.float16)
    qs = torch.randn(12, sq, H, D, generator=g, device=DEV, dtype=torch.float16)
    monkeypatch.setattr(ops, "ATTN_FUSED_MERGE", False)
    want = [ops.attn_decode(qs[i], k, v, sk, scale).clone() for i in range(qs.shape[0])]
    monkeypatch.setattr(ops, "ATTN_FUSED_MERGE", True)
    hog_src = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
    hog_dst = torch.empty_like(hog_src)
    side = torch.cuda.Stream()
    outs = []
    for it in range(600):
        if it % 3 == 0:
            with torch.cuda.stream(side):
                hog_dst.copy_(hog_src)                      # 512 MB of HBM traffic overlapping the next launches
        outs.append(ops.attn_decode(qs[it % qs.shape[0]], k, v, sk, scale))
        if len(outs) == 60:
            torch.cuda.synchronize()
            for j, o in enumerate(outs):
                i = (it - 59 + j) % qs.shape[0]
                assert torch.equal(o, want[i]), f"launch {it - 59 + j}: one-launch merge differs from the two-launch form"
            outs = []
    torch.cuda.synchronize()


@pytest.mark.parametrize("M", [7, 17])
def test_13b_down_proj_takes_the_three_way_split_by_rule(M):
    """The one shape the shipped rule splits across workgroups: 256 < panels <= 384 with K >= 8192 — Llama-2-13B's
    down_proj (N 5120, K 13824: 41.8 -> 34.8 us at 17 rows, profiles/r04_gemm_ksplit_force_ab.jsonl).  Residual + sums of
    squares, both layouts: deterministic, equal to the one-workgroup form up to the re-association of the K sum, and
    against the oracle's fp32-accumulated linear."""
    ops = _ops()
    from triforce_amd import hip
    L = hip.lib()
    N, K = 5120, 13824
    w = rnd(N, K, seed=500, scale=0.02)
    x, res = rnd(M, K, seed=501 + M), rnd(M, N, seed=502)
    pl = ops.PackedLinear(w.to(DEV))
    xd, resd = x.to(DEV), res.to(DEV)

    def run(packed):
        buf = ops.Act.from_rows(resd) if packed else resd.clone()
        ss = ops.ss_buffer(N, DEV)
        ops.linear(ops.Act.from_rows(xd) if packed else xd, pl, resid=buf, out=buf, ss_out=ss)
        return (buf.rows() if packed else buf), ss[:, :M].clone()
    a, sa = run(False)
    b, sb = run(True)
    a2, _ = run(False)
    assert torch.equal(a, b) and torch.equal(sa, sb) and torch.equal(a, a2)
    old = L.tf_sg_tune(4, 1)                                   # never split
    try:
        one, _ = run(False)
    finally:
        L.tf_sg_tune(4, old)
    d = (a.float() - one.float()).abs()
    assert bool((d <= 2 * one.float().abs() * 2 ** -10 + 2e-3).all()) and float((d > 0).float().mean()) < 0.12
    want = res + R.linear(x, w)                                # fp16 residual add of the fp16 GEMM result
    ulp_report("13B down_proj, 3 K-splits across workgroups", a, want, max_ulp_frac=8e-2, ulps=2, atol=2e-3)


def test_set_tokens_writes_ids_positions_and_lengths_in_one_launch():
    """tf_set_tokens: the ids arrive as kernel arguments; padding, positions and the two length scalars of a captured
    forward in the same launch; optional parts left untouched; graph-capturable arguments are by value (a replay repeats
    the captured ids — which is why the decode loop launches it eagerly in front of the replays)."""
    ops = _ops()
    dst = torch.full((1, 9), -7, dtype=torch.long, device=DEV)
    pos = torch.full((1, 7), -7, dtype=torch.long, device=DEV)
    slot = torch.full((1,), -7, dtype=torch.int32, device=DEV)
    sk = torch.full((1,), -7, dtype=torch.int32, device=DEV)
    ops.set_tokens(dst[:, :8], [5, 31999, 0, 2 ** 40 + 3], 100, pos=pos, pos0=124928, slot=slot, sk=sk, sk_val=124936)
    assert dst[0].tolist() == [5, 31999, 0, 2 ** 40 + 3, 100, 100, 100, 100, -7]
    assert pos[0].tolist() == [124928 + i for i in range(7)]
    assert slot.item() == 124928 and sk.item() == 124936
    ops.set_tokens(dst[:, :3], [1], 100)                            # tokens only
    assert dst[0].tolist()[:4] == [1, 100, 100, 2 ** 40 + 3] and pos[0, 0].item() == 124928
    ops.set_tokens(None, (), 0, pos=pos, pos0=3, slot=slot, sk=sk, sk_val=10)   # lengths only
    assert pos[0].tolist() == [3, 4, 5, 6, 7, 8, 9] and slot.item() == 3 and sk.item() == 10 and dst[0, 0].item() == 1
    full = torch.zeros(32, dtype=torch.long, device=DEV)
    ops.set_tokens(full, list(range(1000, 1032)), 100)
    assert full.tolist() == list(range(1000, 1032))
    with pytest.raises(AssertionError):
        ops.set_tokens(torch.zeros(33, dtype=torch.long, device=DEV), [1], 100)


def test_kv_row_copies_for_k_and_v_in_one_launch_match_the_single_tensor_entries():
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(5)
    L, H, T, D = 3, 5, 40, 128
    sk_ = torch.randn(L, H, T, D, generator=g, device=DEV, dtype=torch.float16)
    sv_ = torch.randn(L, H, T, D, generator=g, device=DEV, dtype=torch.float16)
    dk = torch.randn(L, H, 64, D, generator=g, device=DEV, dtype=torch.float16)
    dv = torch.randn(L, H, 64, D, generator=g, device=DEV, dtype=torch.float16)
    wk, wv = dk.clone(), dv.clone()
    ops.kv_copy_rows(sk_, wk, 3, 50, 11)
    ops.kv_copy_rows(sv_, wv, 3, 50, 11)
    ops.kv_copy_rows_pair(sk_, sv_, dk, dv, 3, 50, 11)
    assert torch.equal(dk, wk) and torch.equal(dv, wv)
    ops.kv_copy_rows_pair(sk_[1:2], sv_[1:2], dk[2:3], dv[2:3], 0, 0, 40)        # layer slices (strided views)
    assert torch.equal(dk[2, :, :40], sk_[1]) and torch.equal(dv[2, :, :40], sv_[1]) and torch.equal(dk[:2], wk[:2])
    with pytest.raises(IndexError):
        ops.kv_copy_rows_pair(sk_, sv_, dk, dv, 30, 0, 11)
    # different strides for K and V: falls back to two launches, same result
    sv_wide = torch.randn(L, H, T + 8, D, generator=g, device=DEV, dtype=torch.float16)
    ops.kv_copy_rows_pair(sk_, sv_wide[:, :, :T], dk, dv, 0, 0, 5)
    assert torch.equal(dk[:, :, :5], sk_[:, :, :5]) and torch.equal(dv[:, :, :5], sv_wide[:, :, :5])
    # shifts: overlapping, downwards
    ck, cv = sk_.clone(), sv_.clone()
    rk, rv = sk_.clone(), sv_.clone()
    ops.kv_shift_rows(rk, 9, 2, 31)
    ops.kv_shift_rows(rv, 9, 2, 31)
    ops.kv_shift_rows_pair(ck, cv, 9, 2, 31)
    assert torch.equal(ck, rk) and torch.equal(cv, rv)
    assert torch.equal(ck[:, :, 2:33], sk_[:, :, 9:40]) and torch.equal(cv[:, :, 2:33], sv_[:, :, 9:40])
    with pytest.raises(IndexError):
        ops.kv_shift_rows_pair(ck, cv, 20, 2, 31)
