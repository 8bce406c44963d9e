"""Sequoia tree path on CPU: (1) the oracle restatement (oracle/ref_tree.py) against the golden streams recorded
from the UNMODIFIED reference SpecTree + TP_llama_tree engine (oracle/gen_golden.py `sequoia_case`); (2) the tree
builder against the reference's 512-node fixture; (3) the product's host logic (SpecTree, TP_llama_tree engine,
Sequoia retrieval cache, KV compaction) with the HIP ops swapped for oracle restatements — must reproduce the
oracle's stream token for token under a shared uniform stream.  No kernel is validated here."""
import pytest
import torch

from oracle import ref_model as M
from oracle import ref_tree as RT
from oracle import specs
from tests import helpers as Hh

CASES = ["sequoia_tree512", "sequoia_small"]


def _oracle(g, rng, rand=None):
    tsd = specs.random_state_dict(g["tcfg"], g["tseed"], head_std=g["head_std"])
    gm = RT.grow_map_from_branches(g["branches"])
    eng = RT.TreeEngine(g["tcfg"], tsd, g["prefill"], g["gen_len"], g["budget"], g["chunk"], gm["size"])
    return RT.SpecTreeO(eng, gm, g["temperature"], g["top_p"], g["tcfg"]["vocab_size"], rng, rand), eng, tsd, gm


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_reference_sequoia_stream(name):
    g = Hh.load_golden(name)
    torch.manual_seed(g["rng_seed"])
    so, eng, _, gm = _oracle(g, M.TorchRng())
    prompt = Hh.prompt_of(g)[0]
    gen, counts = RT.run_sequoia(so, prompt, g["gen_len"])
    assert gen == g["generated"]
    assert counts == [s["acc_count"] for s in g["steps"] if not s["terminal"]]
    for tr, s in zip(so.trace, g["steps"]):
        assert torch.equal(tr["tokens"], s["tree_tokens"])
        if not s["terminal"]:
            assert tr["tokens"][tr["accept_list"]].tolist() + [tr["next_token"]] == s["accept_tokens"]
    assert eng.kv_cache.seq_len == g["final_seq_len"]
    assert torch.equal(gm["mask"].sum(dim=1), g["mask_rowsum"]) and torch.equal(gm["depth"], g["depth"])


def test_tree_builder_matches_reference_fixture_and_oracle():
    from triforce_amd.utils import tree as T
    g = Hh.load_golden("sequoia_tree512")                # branches of the reference's tree/512.pt
    ours, theirs = T.grow_map_from_branches(g["branches"]), RT.grow_map_from_branches(g["branches"])
    for k in ("roots", "branches", "Successors", "size"):
        assert ours[k] == theirs[k]
    assert torch.equal(ours["mask"], theirs["mask"]) and torch.equal(ours["depth"], theirs["depth"])
    assert ours["size"] == 512 and int(ours["depth"].max()) == 15 and torch.equal(ours["mask"].sum(1), g["mask_rowsum"])
    off, flat = T.successors_csr(ours["Successors"])
    assert off[-1] == 511 and flat.tolist() == list(range(1, 512))         # children are numbered level by level
    # the dynamic program: value of the expanded tree == the DP optimum, more nodes never hurt, depth cap respected
    F, back = T.search_tree(T.DEFAULT_ACCEPTANCE, 96, 10)
    gm = T.grow_map_from_branches(T.expand_tree(back, 96, 10))
    assert gm["size"] == 96 and int(gm["depth"].max()) <= 9

    def value(node):
        return 1 + sum(T.DEFAULT_ACCEPTANCE[i + 1] * value(k) for i, k in enumerate(gm["Successors"][node]))
    assert abs(value(0) - F[96, 10]) < 1e-9
    assert all(F[m + 1, 10] >= F[m, 10] - 1e-12 for m in range(1, 95))
    # with the published acceptance vector our DP lands on the same level widths as the reference's 512-node tree
    assert [len(r) for r in T.load_grow_map("512")["roots"]] == [len(r) for r in ours["roots"]]


def test_pack_tree_mask_roundtrip():
    from tests import cpu_backend
    from triforce_amd import ops
    m = (torch.rand(37, 100, generator=torch.Generator().manual_seed(1)) > 0.6)
    bits = ops.pack_tree_mask(m)
    assert bits.dtype == torch.int32 and bits.shape == (37, 4)
    assert torch.equal(cpu_backend._unpack_bits(bits, 100), m)


@pytest.mark.parametrize("name", CASES)
def test_product_host_logic_reproduces_oracle_stream(name, cpu_ops):
    """SpecTree + TP_llama_tree.DistributedLlama + DistributedRetrievalCache_Seqouia + gather_kv_incremental on
    CPU (ops patched): same uniform stream and rand table as the oracle -> same tokens, same accept counts,
    same compacted KV."""
    from triforce_amd.models.config_yarn import LlamaConfig
    from triforce_amd.models.TP_llama_tree import DistributedLlama
    from triforce_amd.utils.SpecTree_TP import SpecTree
    from triforce_amd.utils.sampling import UniformSource
    from triforce_amd.utils.tree import grow_map_from_branches
    g = Hh.load_golden(name)
    V = g["tcfg"]["vocab_size"]
    uniforms = Hh.fixed_uniforms(4096, seed=7)
    gm = grow_map_from_branches(g["branches"])
    rand = torch.rand(gm["size"], V, generator=torch.Generator().manual_seed(3)).half()
    so, oeng, tsd, _ = _oracle(g, M.InjectedRng(uniforms), rand=rand)
    prompt = Hh.prompt_of(g)[0]
    want, want_counts = RT.run_sequoia(so, prompt, g["gen_len"])

    cfg = LlamaConfig.from_dict(g["tcfg"])
    llm = DistributedLlama("unused", config=cfg, device="cpu", local_rank=0, world_size=1, prefill=g["prefill"],
                           gen_len=g["gen_len"], temperature=g["temperature"], top_p=g["top_p"],
                           retrieval_budget=g["budget"], retrieval_chunk_size=g["chunk"], kv_offload=True,
                           on_chip_layers=cfg.num_hidden_layers, tree_size=gm["size"])
    llm.init_parameters(tsd)
    st = SpecTree(llm, temperature=g["temperature"], top_p=g["top_p"], max_length=g["prefill"] + g["gen_len"],
                  vocab_size=V, grow_map=gm, rng=UniformSource("cpu", values=uniforms), rand_values=rand)
    next_token = st.prefill(prompt)
    got, counts, n = [int(next_token)], [], 0
    while n < g["gen_len"]:
        st.construct_grow_map(next_token)
        next_token, acc, toks = st.verify()
        if next_token is None:
            break
        got.extend(toks[1:].tolist())
        n += acc
        counts.append(acc)
        next_token = next_token.unsqueeze(0)
    assert got == want and counts == want_counts
    S = llm.kv_cache.seq_len
    assert S == oeng.kv_cache.seq_len
    # compacted full cache and refreshed retrieval tail agree with the oracle's (head-major vs token-major views)
    assert torch.equal(llm.kv_cache.k[:, :, g["prefill"]:S].permute(0, 2, 1, 3), oeng.kv_cache.key_cache[:, g["prefill"]:S])
    gN = S - g["prefill"]
    B = g["budget"]
    assert torch.equal(llm.retrieval_cache.k[:, :, B - gN:B].permute(0, 2, 1, 3),
                       oeng.retrieval_cache.key_cache[:, B - gN:B])


def test_gather_kv_incremental_rejects_unsorted_and_handles_identity(cpu_ops):
    from triforce_amd.models.cache import DistributedSimpleCache
    from triforce_amd.models.config_yarn import LlamaConfig
    from triforce_amd.models.TP_layers import DistributedOffloadingConfig
    cfg = DistributedOffloadingConfig(LlamaConfig.from_dict(specs.tiny_target_config(vocab_size=64)), 0, 1)
    c = DistributedSimpleCache(cfg, max_budget=64, device="cpu", on_chip_layers=cfg.num_hidden_layers)
    c.k.normal_()
    c.v.normal_()
    k0 = c.k.clone()
    c.gather_kv_incremental([0, 1, 2], 10)               # identity prefix: nothing moves
    assert c.seq_len == 13 and torch.equal(c.k, k0)
    c.gather_kv_incremental([0, 3, 7, 20], 10)
    assert c.seq_len == 14 and torch.equal(c.k[:, :, 10:14], k0[:, :, [10, 13, 17, 30]])
    with pytest.raises(AssertionError):
        c.gather_kv_incremental([0, 5, 3], 10)


def test_tree_builder_agrees_with_oracle_on_random_trees():
    """grow_map_from_branches (product, utils/tree.py) vs the oracle's builder (pinned to the reference's tree/512.pt)
    on 40 random ragged trees — zero-child nodes, single chains, wide levels — including the CSR successor table the
    accept-walk kernel consumes and the packed mask the attention kernel consumes."""
    import random
    from triforce_amd.utils import tree as T
    rnd = random.Random(5)
    for _ in range(40):
        branches, width = [], 1
        for _level in range(rnd.randint(1, 6)):
            level = [rnd.choice([0, 0, 1, 1, 2, 3, 5]) for _ in range(width)]
            if sum(level) == 0:
                level[rnd.randrange(width)] = 1
            branches.append(level)
            width = sum(level)
            if width > 40:
                break
        ours, theirs = T.grow_map_from_branches(branches), RT.grow_map_from_branches(branches)
        for k in ("roots", "branches", "Successors", "size"):
            assert ours[k] == theirs[k], (k, branches)
        assert torch.equal(ours["mask"], theirs["mask"]) and torch.equal(ours["depth"], theirs["depth"])
        n = ours["size"]
        off, flat = T.successors_csr(ours["Successors"])
        assert off.tolist()[0] == 0 and off.tolist()[-1] == n - 1 and sorted(flat.tolist()) == list(range(1, n))
        for node in range(n):                             # a node sees exactly itself and its ancestors
            row = ours["mask"][node]
            seen, cur = {node}, node
            parents = {c: p for p in range(n) for c in ours["Successors"][p]}
            while cur in parents:
                cur = parents[cur]
                seen.add(cur)
            assert set(torch.nonzero(row).reshape(-1).tolist()) == seen
            assert int(ours["depth"][node]) == len(seen) - 1
