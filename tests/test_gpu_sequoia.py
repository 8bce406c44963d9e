"""Sequoia tree path on a real MI355X, through the C ABI: tree-masked block attention (tf_attn_block), the
device-side tree walk (tf_tree_accept), KV compaction (tf_kv_gather_rows) and the whole SpecTree loop of the
product (models/TP_llama_tree.py + utils/SpecTree_TP.py) against the CPU oracle (oracle/ref_tree.py), which is
itself pinned token-for-token against the unmodified reference (tests/golden/sequoia_*.pt)."""
import math

import pytest
import torch

from oracle import ref_model as M
from oracle import ref_ops as R
from oracle import ref_tree as RT
from oracle import specs
from tests import helpers as Hh
from tests.test_gpu_e2e import _logit_check
from tests.test_gpu_ops import ATTN_ATOL, ATTN_RTOL, _attn_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _random_tree(n, seed, max_kids=4):
    g = torch.Generator().manual_seed(seed)
    branches, level, total = [], 1, 1
    while total < n:
        blist = []
        for _ in range(level):
            b = min(int(torch.randint(0, max_kids + 1, (1,), generator=g)), n - total)
            blist.append(b)
            total += b
        if sum(blist) == 0:
            blist[0] = min(2, n - total)
            total += blist[0]
        branches.append(blist)
        level = sum(blist)
    return RT.grow_map_from_branches(branches)


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sq,prefix,T,H,D,row0", [
    (1, 128, 64, 2, 128, 0), (7, 300, 64, 4, 128, 1), (53, 4096, 512, 4, 128, 406), (33, 1003, 100, 2, 64, 5),
    (128, 2000, 512, 2, 128, 200), (64, 0, 96, 2, 128, 0), (16, 12288, 512, 2, 128, 300)])
def test_tree_attention_matches_sdpa_oracle(sq, prefix, T, H, D, row0):
    """q rows = tree nodes row0..row0+sq of a random tree; keys = `prefix` unmasked rows + T tree slots."""
    from triforce_amd import ops
    gm = _random_tree(T, seed=sq + T)
    sk = prefix + T
    q, k, v, kd, vd = _attn_inputs(sq, sk, H, D, seed=50 + sq)
    bits = ops.pack_tree_mask(gm["mask"].to(DEV))
    add = torch.cat([torch.zeros(sq, prefix), RT.additive_tree_mask(gm["mask"])[row0:row0 + sq].float()], dim=-1)
    want = RT.attn_sdpa(q, k, v, add).reshape(sq, H * D)
    got = ops.attn_tree(q.to(DEV), kd, vd, sk, 1.0 / math.sqrt(D), bits, prefix, mask_row0=row0)
    Hh.close(got.float().cpu(), want.float(), atol=ATTN_ATOL, rtol=ATTN_RTOL)
    for ns in (1, 3):                                     # split count must not matter
        got2 = ops.attn_block(q.to(DEV), kd, vd, sk, 1.0 / math.sqrt(D), nsplit=ns, tree_mask=bits, mask_row0=row0,
                              tree_start=prefix)
        Hh.close(got2.float(), got.float(), atol=ATTN_ATOL, rtol=ATTN_RTOL)


def test_tree_attention_verify_shape_512_rows():
    """The verify pass: all 512 tree nodes as queries (4 slabs of 128) over prefix + 512 tree keys."""
    from triforce_amd import ops
    g = Hh.load_golden("sequoia_tree512")
    gm = RT.grow_map_from_branches(g["branches"])
    sq, prefix, H, D = 512, 3001, 2, 128
    q, k, v, kd, vd = _attn_inputs(sq, prefix + 512, H, D, seed=9)
    bits = ops.pack_tree_mask(gm["mask"].to(DEV))
    add = torch.cat([torch.zeros(sq, prefix), RT.additive_tree_mask(gm["mask"]).float()], dim=-1)
    want = RT.attn_sdpa(q, k, v, add).reshape(sq, H * D)
    got = ops.attn_tree(q.to(DEV), kd, vd, prefix + 512, 1.0 / math.sqrt(D), bits, prefix)
    Hh.close(got.float().cpu(), want.float(), atol=2.5e-4, rtol=ATTN_RTOL)     # 512 rows: measured max 6.1e-5
    # a node must not see anything but its ancestors: perturbing a non-ancestor key/value leaves its row unchanged
    node = 300
    stranger = next(j for j in range(1, 512) if gm["mask"][node, j] == 0)
    kd2, vd2 = kd.clone(), vd.clone()
    kd2[:, prefix + stranger] += 3.0
    vd2[:, prefix + stranger] -= 5.0
    got3 = ops.attn_tree(q.to(DEV), kd2, vd2, prefix + 512, 1.0 / math.sqrt(D), bits, prefix)
    assert torch.equal(got3[node], got[node])


# ------------------------------------------------------------------------------------------
def _walk_case(seed, V=1000, n=60, T=0.6, agree=4.0, eos_token=None, identical=False):
    g = torch.Generator().manual_seed(seed)
    gm = _random_tree(n, seed=seed)
    N = gm["size"]
    draft = torch.randn(N, V, generator=g) * 2.0
    target_logits = draft + torch.randn(N, V, generator=g) / agree
    p = torch.softmax(target_logits / T, dim=-1)
    if identical:                                         # p == q everywhere: every rejection leaves a 0/0 residual
        p = torch.softmax(draft / T, dim=-1)
    tokens = torch.zeros(N, dtype=torch.long)
    for node in range(N):                                 # children drawn without replacement from the draft row
        kids = gm["Successors"][node]
        if kids:
            tokens[kids] = RT.sample_without_replacement(draft[node:node + 1], torch.rand(1, V, generator=g).half(),
                                                         len(kids), T) + (3 if eos_token is None else 0)
    tokens = tokens.clamp_(max=V - 1)
    if eos_token is not None:
        tokens[gm["Successors"][0][0]] = eos_token
    u = torch.rand(300, generator=g)
    return gm, p, draft, tokens, u


@pytest.mark.parametrize("seed", list(range(24)))
def test_tree_accept_walk_bit_exact(seed):
    """Same p / draft logits / tokens / uniforms -> identical accept list, next token, terminal flag, consumption."""
    from triforce_amd import ops
    from triforce_amd.utils.tree import successors_csr
    gm, p, draft, tokens, u = _walk_case(seed, agree=[0.5, 2.0, 8.0][seed % 3])
    acc, nxt, terminal, count = RT.accept_walk(p, draft, tokens, gm["Successors"], 0.6, M.InjectedRng(u.tolist()))
    off, flat = successors_csr(gm["Successors"], DEV)
    out = torch.zeros(ops.TREE_ACCEPT_OUT, dtype=torch.int64, device=DEV)
    ops.tree_accept(p.to(DEV), draft.to(DEV), tokens.to(DEV), off, flat, u.to(DEV), 0.6, out)
    rec = out.tolist()
    assert rec[0] == len(acc) and rec[4:4 + rec[0]] == acc, (rec[:12], acc)
    assert bool(rec[2]) == terminal
    if not terminal:
        assert rec[1] == nxt
    r2 = M.InjectedRng(u.tolist())                         # uniforms consumed: one per examined child (+1 for the sample)
    RT.accept_walk(p, draft, tokens, gm["Successors"], 0.6, r2)
    assert rec[3] == r2.i and count == len(acc) - 1 + (0 if terminal else 1)


def test_tree_accept_edge_cases():
    from triforce_amd import ops
    from triforce_amd.utils.tree import successors_csr

    def run(gm, p, draft, tokens, u):
        off, flat = successors_csr(gm["Successors"], DEV)
        out = torch.zeros(ops.TREE_ACCEPT_OUT, dtype=torch.int64, device=DEV)
        ops.tree_accept(p.to(DEV), draft.to(DEV), tokens.to(DEV), off, flat, u.to(DEV), 0.6, out)
        return out.tolist()
    # (a) single-node tree: the root is a leaf -> bonus token from the target row, one uniform consumed
    gm = RT.grow_map_from_branches([[0]])
    p = torch.softmax(torch.randn(1, 500, generator=torch.Generator().manual_seed(1)), dim=-1)
    rec = run(gm, p, torch.zeros(1, 500), torch.zeros(1, dtype=torch.long), torch.tensor([0.37] * 8))
    assert rec[0] == 1 and rec[2] == 0 and rec[3] == 1 and rec[1] == R.sample_inverse_cdf(p[0], 0.37)
    # (b) an accepted token 2 (eos) is terminal (SpecTree_TP.py:188-190): no sample is drawn
    gm, p, draft, tokens, u = _walk_case(3, eos_token=2)
    p[0, 2] = 1.0                                          # p[tok] > r*q[tok] for any r < 1
    rec = run(gm, p, draft, tokens, u)
    acc, nxt, terminal, _ = RT.accept_walk(p, draft, tokens, gm["Successors"], 0.6, M.InjectedRng(u.tolist()))
    assert terminal and rec[2] == 1 and rec[0] == len(acc) == 2 and rec[3] == 1
    # (c) a NaN target row: every test `p[tok] > r*q[tok]` is False, the residual stays NaN -> terminal (:199-200)
    gm, p, draft, tokens, u = _walk_case(5)
    p[0, :] = float("nan")
    rec = run(gm, p, draft, tokens, u)
    acc, nxt, terminal, _ = RT.accept_walk(p, draft, tokens, gm["Successors"], 0.6, M.InjectedRng(u.tolist()))
    assert terminal and acc == [0] and rec[2] == 1 and rec[0] == 1 and rec[3] == len(gm["Successors"][0])


@pytest.mark.parametrize("rows,V,k,T", [(1, 1024, 4, 0.6), (7, 32000, 7, 0.6), (53, 32000, 5, 0.8), (31, 5000, 16, 1.0)])
def test_sample_without_replacement_matches_oracle(rows, V, k, T):
    """tf_sample_without_replacement == (rand.log() / softmax(logits / T)).topk(k).indices, row by row and in order
    (a winner may differ only where two keys are within fp32 rounding of each other)."""
    from triforce_amd import ops
    g = torch.Generator().manual_seed(rows + V)
    logits = torch.randn(rows, V, generator=g) * 2.0
    rand = torch.rand(rows, V, generator=g).half()
    rand[0, :3] = 0.0                                      # log(0) = -inf: never drawn
    want = RT.sample_without_replacement(logits, rand, k, T).view(rows, k)
    got = ops.sample_without_replacement(logits.to(DEV), rand.to(DEV), k, T).view(rows, k).cpu()
    # fp16 uniforms round to exactly 1.0 now and then (log = 0 -> key 0, the best possible): with 32 000 columns a
    # row holds several such ties and torch.topk's order among equal keys is unspecified (ours: lowest token id).
    # The race KEYS of the k winners must match position by position; the ids must where keys are distinct.
    for r in range(rows):
        assert len(set(got[r].tolist())) == k and int(got[r].min()) >= 0 and int(got[r].max()) < V
        q = torch.softmax(logits[r] / T, -1)
        keys = rand[r].log().float() / q
        Hh.close(keys[got[r]], keys[want[r]], rtol=1e-4, atol=0.0)
        distinct = keys[want[r]].unique().numel() == k and (keys == keys[want[r]][-1]).sum() == 1
        if distinct:
            differ = (got[r] != want[r]).nonzero().flatten().tolist()
            for j in differ:                                # only a near-tie may swap two neighbours
                assert abs(float(keys[got[r][j]] / keys[want[r][j]]) - 1) < 1e-5
        ties = keys[got[r]][keys[got[r]] == 0]
        if ties.numel():                                   # among tied zeros: ascending token ids
            z = got[r][keys[got[r]] == 0].tolist()
            assert z == sorted(z)
    assert not (got[0].unsqueeze(1) == torch.arange(3).unsqueeze(0)).any()


def test_kv_gather_rows_bit_exact():
    from triforce_amd import ops
    L, H, T, D = 3, 4, 300, 128
    g = torch.Generator().manual_seed(2)
    k = torch.randn(L, H, T, D, generator=g).half().to(DEV)
    v = torch.randn(L, H, T, D, generator=g).half().to(DEV)
    k0, v0 = k.clone(), v.clone()
    idx = [0, 1, 5, 6, 40, 41, 200, 201, 202, 230, 231, 232, 233, 234, 235, 236, 237, 250]      # > one 16-row batch
    ops.kv_gather_rows(k, v, 37, torch.tensor(idx, dtype=torch.int32, device=DEV))
    src = [37 + i for i in idx]
    assert torch.equal(k[:, :, 37:37 + len(idx)], k0[:, :, src]) and torch.equal(v[:, :, 37:37 + len(idx)], v0[:, :, src])
    assert torch.equal(k[:, :, :37], k0[:, :, :37]) and torch.equal(k[:, :, 37 + len(idx) + 250:], k0[:, :, 37 + len(idx) + 250:])


# ------------------------------------------------------------------------------------------
def _product(g, gm, tsd, uniforms=None, rand=None, on_chip=None):
    from triforce_amd.models.config_yarn import LlamaConfig
    from triforce_amd.models.TP_llama_tree import DistributedLlama
    from triforce_amd.utils.SpecTree_TP import SpecTree
    from triforce_amd.utils.sampling import UniformSource
    cfg = LlamaConfig.from_dict(g["tcfg"])
    llm = DistributedLlama("unused", config=cfg, device=DEV, local_rank=0, world_size=1, prefill=g["prefill"],
                           gen_len=g["gen_len"], temperature=g["temperature"], top_p=g["top_p"],
                           retrieval_budget=g["budget"], retrieval_chunk_size=g["chunk"], kv_offload=True,
                           on_chip_layers=cfg.num_hidden_layers if on_chip is None else on_chip, tree_size=gm["size"])
    llm.init_parameters(tsd)
    rng = UniformSource(DEV, values=uniforms) if uniforms is not None else UniformSource(DEV, seed=3)
    st = SpecTree(llm, temperature=g["temperature"], top_p=g["top_p"], max_length=g["prefill"] + g["gen_len"],
                  vocab_size=g["tcfg"]["vocab_size"], grow_map=gm, rng=rng, rand_values=rand)
    return llm, st


@pytest.mark.parametrize("name", ["sequoia_tree512", "sequoia_small"])
def test_tree_forward_logits_teacher_forced(name):
    """Stage-wise parity with the ORACLE's tree tokens forced into the product: per-level retrieval-cache logits,
    then the 512-row target verify logits, then the probabilities that feed the accept walk."""
    from triforce_amd.models.TP_llama import TreeMask
    from triforce_amd.utils.sampling import norm_logits
    g = Hh.load_golden(name)
    gm = RT.grow_map_from_branches(g["branches"])
    V = g["tcfg"]["vocab_size"]
    uniforms = Hh.fixed_uniforms(4096, seed=7)
    rand = torch.rand(gm["size"], V, generator=torch.Generator().manual_seed(3)).half()
    tsd = specs.random_state_dict(g["tcfg"], g["tseed"], head_std=g["head_std"])
    oeng = RT.TreeEngine(g["tcfg"], tsd, g["prefill"], g["gen_len"], g["budget"], g["chunk"], gm["size"])
    so = RT.SpecTreeO(oeng, gm, g["temperature"], g["top_p"], V, M.InjectedRng(uniforms), rand)
    prompt = Hh.prompt_of(g)[0]
    first = so.prefill(prompt)
    so.construct_grow_map(first)
    llm, st = _product(g, gm, tsd, uniforms, rand)
    st.prefill(prompt.to(DEV))
    S = llm.kv_cache.seq_len
    assert S == oeng.kv_cache.seq_len
    # retrieval cache built from the same prefix: same chunk choice up to score ties -> compare the logits it yields
    toks = so.verify_tokens.to(DEV)
    B = g["budget"]
    level = [(0, 1)] + [(st.level_start[i], st.level_start[i] + sum(st.branches[i])) for i in range(st.draft_step - 1)]
    for (a, b) in level:
        lg = llm.retrieval_tree_inference(input_ids=toks[a:b].view(1, -1), storage_ids=range(B + a, B + b),
                                          position_ids=(st.depth[a:b] + S).unsqueeze(0),
                                          attention_mask=TreeMask(st.mask_bits, a))[0]
        # a node attends to the device-computed KV rows of up to 15 ancestors (each within fp16 noise of the oracle's):
        # 3 fp16 spacings at the largest logit instead of the 2 of a single forward
        _logit_check(f"{name} tree level [{a},{b})", lg.float().cpu(), so.draft_logits[a:b], spacings=3.0)
    # the dense-mask form of the reference API gives the same result as the packed form
    a, b = level[2]
    dense = torch.cat([torch.zeros(b - a, B), RT.additive_tree_mask(gm["mask"])[a:b].float()], dim=-1).half()
    lg2 = llm.retrieval_tree_inference(input_ids=toks[a:b].view(1, -1), storage_ids=torch.arange(B + a, B + b),
                                       position_ids=(st.depth[a:b] + S).unsqueeze(0),
                                       attention_mask=dense[None, None].to(DEV))[0]
    _logit_check(f"{name} dense-mask API", lg2.float().cpu(), so.draft_logits[a:b], spacings=3.0)
    # target verify over the full cache with the tree mask
    pos = (gm["depth"] + S).unsqueeze(0)
    add = torch.cat([torch.zeros(gm["size"], S), so.tree_mask.float()], dim=-1)
    want = oeng.inference(so.verify_tokens.unsqueeze(0), position_ids=pos, attention_mask=add)[0]
    got = llm.inference(input_ids=toks.unsqueeze(0), position_ids=pos.to(DEV), attention_mask=TreeMask(st.mask_bits, 0))[0]
    _logit_check(f"{name} tree verify", got.float().cpu(), want, spacings=3.0)
    pw = R.norm_logits(want, g["temperature"], -1, g["top_p"])
    pg = norm_logits(got, temperature=g["temperature"], top_k=-1, top_p=g["top_p"]).cpu()
    assert (pw - pg).abs().max() < 2e-2 and ((pw > 0) != (pg > 0)).float().mean() < 2e-3


def test_captured_tree_growth_equals_eager():
    """The whole tree growth (15 sampling steps + 16 level forwards) as ONE hipGraph == the eager level loop."""
    g = Hh.load_golden("sequoia_tree512")
    gm = RT.grow_map_from_branches(g["branches"])
    V = g["tcfg"]["vocab_size"]
    rand = torch.rand(gm["size"], V, generator=torch.Generator().manual_seed(3)).half()
    tsd = specs.random_state_dict(g["tcfg"], g["tseed"], head_std=g["head_std"])
    llm, st = _product(g, gm, tsd, Hh.fixed_uniforms(4096, seed=7), rand)
    nxt = st.prefill(Hh.prompt_of(g)[0].to(DEV))
    st.construct_grow_map(nxt)
    toks, logits = st.verify_tokens.clone(), st.draft_logits.clone()
    tree_k = llm.retrieval_cache.k[:, :, st.storage0:].clone()
    assert st.capture_grow_graph()
    assert int(st.verify_tokens.abs().sum()) == 0
    st.construct_grow_map(nxt)
    assert torch.equal(st.verify_tokens, toks) and torch.equal(st.draft_logits, logits)
    assert torch.equal(llm.retrieval_cache.k[:, :, st.storage0:], tree_k)
    # a later step (other root token, longer cache) through the same graph == eager
    nt, acc, _ = st.verify()
    assert nt is not None
    st.construct_grow_map(nt.unsqueeze(0))
    toks2, logits2 = st.verify_tokens.clone(), st.draft_logits.clone()
    st._grow_graph = None
    st.construct_grow_map(nt.unsqueeze(0))
    assert torch.equal(st.verify_tokens, toks2) and torch.equal(st.draft_logits, logits2)


@pytest.mark.parametrize("name,on_chip", [("sequoia_tree512", None), ("sequoia_small", None), ("sequoia_small", 1)])
def test_spectree_loop_on_device(name, on_chip):
    """Free-running product loop (with host-offloaded layers in the last case): structural invariants at every
    step and a common prefix with the oracle stream under the shared uniform stream / rand table."""
    g = Hh.load_golden(name)
    gm = RT.grow_map_from_branches(g["branches"])
    V = g["tcfg"]["vocab_size"]
    uniforms = Hh.fixed_uniforms(4096, seed=7)
    rand = torch.rand(gm["size"], V, generator=torch.Generator().manual_seed(3)).half()
    tsd = specs.random_state_dict(g["tcfg"], g["tseed"], head_std=g["head_std"])
    oeng = RT.TreeEngine(g["tcfg"], tsd, g["prefill"], g["gen_len"], g["budget"], g["chunk"], gm["size"])
    so = RT.SpecTreeO(oeng, gm, g["temperature"], g["top_p"], V, M.InjectedRng(uniforms), rand)
    prompt = Hh.prompt_of(g)[0]
    want, want_counts = RT.run_sequoia(so, prompt, g["gen_len"])

    llm, st = _product(g, gm, tsd, uniforms, rand, on_chip=on_chip)
    if name == "sequoia_tree512":
        assert st.capture_grow_graph()                    # this case runs the captured tree growth
    next_token = st.prefill(prompt.to(DEV))
    got, counts, n = [int(next_token)], [], 0
    P = g["prefill"]
    while n < g["gen_len"]:
        S0 = llm.kv_cache.seq_len
        st.construct_grow_map(next_token)
        tree_tokens = st.verify_tokens.clone()
        assert int(tree_tokens[0]) == got[-1] and int(tree_tokens.min()) >= 0 and int(tree_tokens.max()) < V
        for node in range(gm["size"]):                    # siblings are drawn WITHOUT replacement
            kids = gm["Successors"][node]
            assert len(set(tree_tokens[kids].tolist())) == len(kids)
        next_token, acc, toks = st.verify()
        if next_token is None:
            break
        assert llm.kv_cache.seq_len == S0 + acc and toks.numel() == acc + 1 and int(toks[0]) == got[-1]
        got.extend(toks[1:].tolist())
        n += acc
        counts.append(acc)
        next_token = next_token.unsqueeze(0)
        # retrieval tail == generated rows of the full cache (layer 0 is on chip in every case)
        gN = llm.kv_cache.seq_len - P
        B = g["budget"]
        assert torch.equal(llm.retrieval_cache.k[0, :, B - gN:B], llm.kv_cache.k[0, :, P:P + gN])
    assert n >= min(g["gen_len"], 8)
    cp = Hh.common_prefix(got, want)
    assert cp >= 1, (cp, got[:16], want[:16])
