"""Generate tests/golden/*.pt by running the UNMODIFIED reference (/root/reference) on CPU.

TEST INFRASTRUCTURE ONLY.  Run in the build container (the reference tree is absent on the
GPU box):   python -m oracle.gen_golden [sequoia | sequoia2 | cli | tp | tp2 | shards | offloading]   (no argument = everything)

  (default)   rope_tables, forward_small, cfg1_greedy, cfg1_stochastic, small_gamma6   on-chip path (test/on_chip.py)
  sequoia     sequoia_tree512, sequoia_small            SpecTree + TP_llama_tree (test/offloading_seqouia.py)
  tp          tp_chain                                  TP_llama + TriForce_Dist at world size 1 (test/offloading_TP.py)
  tp2         tp_world2, tp_world4                      the same engine as 2 / 4 gloo processes: shards + all-reduces
  tp8         tp_world8                                 ... as 8 gloo processes (8 heads, one per rank): BASELINE configs[4]'s world size
  shards      tp_shards                                 TP_layers' own weight slicing for every rank of a 4- / 8-way split
  sequoia2    sequoia_world2                            the Sequoia loop as TWO gloo processes
  offloading  offloading_small                          OffloadingFlashSimpleCache (test/offloading.py)
  cli         cli_flags                                 the four scripts' command lines
While generating, every case is also replayed through the CPU restatement (oracle/ref_ops.py,
oracle/ref_model.py) and must agree (logits bit-identical, token streams identical, top-k
tie-tolerant) — that is how the oracle is pinned (the reference has no
golden vectors of its own for this path).
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import _refshim, specs, ref_ops as R, ref_model as M  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def build_reference_model(ref, cfg, sd, draft=False):
    """Constructor + load_state_dict (never from_pretrained: SURVEY §8c pitfall (i))."""
    kw = {k: v for k, v in cfg.items() if not k.startswith("_")}
    hf_cfg = ref.config_yarn.LlamaConfig(**kw)
    hf_cfg._name_or_path = cfg["_name_or_path"]
    cls = ref.modeling_llama_68m.LlamaForCausalLM if draft else ref.modeling_llama.LlamaForCausalLM
    m = cls(hf_cfg)
    m = m.to(torch.float16)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("rotary_emb" in k for k in missing), missing
    return m.eval()


def triforce_case(name, tcfg, dcfg, tseed, dseed, pseed, prefill, budget, gamma, chunk, gen_len,
                  temperature, top_p, rng_seed=None, repeats=1, head_std=0.05):
    ref = _refshim.load_reference()
    tsd = specs.random_state_dict(tcfg, tseed, head_std=head_std)
    dsd = specs.random_state_dict(dcfg, dseed, head_std=head_std)
    prompt = specs.random_prompt(tcfg["vocab_size"], prefill, pseed)
    recent = 256 - 16 - gamma                                    # on_chip.py:76-80

    # ---------------- reference ----------------
    target = build_reference_model(ref, tcfg, tsd)
    draft = build_reference_model(ref, dcfg, dsd, draft=True)
    cache = ref.cache.FlashSimpleCache(target, prefill + gen_len + 16)
    gcache = ref.cache.RetrievalCache(target, max_budget=budget, prefill=prefill, gamma=gamma, chunk_size=chunk)
    dcache = ref.cache.StreamingLLMEvictionCache(draft, start_size=16, recent_size=recent, gamma=gamma)
    eng = _refshim.EagerEngine(ref, target, cache, gcache, draft, dcache, temperature, top_p)
    tok = _refshim.FakeTokenizer()

    # record the retrieval build (scores fed to torch.topk and the indices it returned)
    topk_log = []
    real_topk = torch.topk

    def spy_topk(x, k, dim=-1, **kw):
        out = real_topk(x, k=k, dim=dim, **kw)
        if x.dim() == 3 and x.dtype == torch.float16:
            topk_log.append((x.clone(), out[1].clone()))
        return out

    streams = []
    ref_dec = ref.decoding
    emitted = []
    orig_stream = ref_dec.spec_stream

    def record(t, tokenizer, color="blue"):
        emitted.append(int(torch.as_tensor(t).reshape(-1)[0]))

    ref_dec.spec_stream = record

    def run_ref_triforce():
        emitted.clear()
        if rng_seed is not None:
            torch.manual_seed(rng_seed)
        torch.topk = spy_topk
        try:
            acc, _ = ref_dec.TriForce(tok, eng, prompt, gamma=gamma, max_len=gen_len, top_k=-1, top_p=top_p,
                                      temperature=temperature, verbose=True)
        finally:
            torch.topk = real_topk
        return list(emitted), acc

    def run_ref_ar():
        emitted.clear()
        if rng_seed is not None:
            torch.manual_seed(rng_seed)
        ref_dec.Autoregressive(tok, eng, prompt, max_len=gen_len, top_k=-1, top_p=top_p,
                               temperature=temperature, verbose=True)
        return list(emitted)

    t0 = time.time()
    ar_tokens = run_ref_ar()
    for _ in range(repeats):
        topk_log.clear()
        toks, acc = run_ref_triforce()
        streams.append(dict(tokens=toks, acceptance_rate=acc,
                            final_seq_len=cache.seq_len, draft_seq_len=dcache.seq_len))
    ref_dec.spec_stream = orig_stream
    t_ref = time.time() - t0
    ref_scores = torch.stack([s[0][0] for s in topk_log])       # (L, H, C-1) fp16 (chunk 0 dropped)
    ref_topk_rest = torch.stack([s[1][0] for s in topk_log])     # (L, H, sets-1) (0-based into [:,1:])
    ref_retr_k = gcache.key_cache[:, 0].clone()
    ref_retr_v = gcache.value_cache[:, 0].clone()

    # ---------------- oracle restatement ----------------
    ot, od = M.OracleTarget(tcfg, tsd), M.OracleDraft(dcfg, dsd)
    okv = M.FullCache(tcfg, prefill + gen_len + 16)
    ogc = M.RetrievalCacheO(tcfg, budget, prefill, chunk, gamma)
    odc = M.StreamingCacheO(dcfg, gamma=gamma, start_size=16, recent_size=recent)
    oeng = M.OracleEngine(ot, okv, ogc, od, odc, temperature, top_p)
    if rng_seed is not None:
        torch.manual_seed(rng_seed)
    o_ar = M.autoregressive(oeng, prompt, gen_len, temperature, top_p)
    assert o_ar == ar_tokens, f"[{name}] AR stream mismatch\n{o_ar}\n{ar_tokens}"
    o_streams = []
    for rep in range(repeats):
        if rng_seed is not None:
            torch.manual_seed(rng_seed)
        trace = []
        res = M.triforce(oeng, prompt, gamma, gen_len, temperature, top_p, trace=trace)
        o_streams.append(res)
        assert res["tokens"] == streams[rep]["tokens"], \
            f"[{name}] TriForce stream mismatch (rep {rep})\n{res['tokens']}\n{streams[rep]['tokens']}"
        assert abs(res["acceptance_rate"] - streams[rep]["acceptance_rate"]) < 1e-12
        assert okv.seq_len == streams[rep]["final_seq_len"] and odc.seq_len == streams[rep]["draft_seq_len"]
    # stage-wise: scores bit-identical, top-k tie-tolerant, and (given the canonical tie rule)
    # gathered cache identical wherever the index sets agree
    o_scores = torch.stack(ogc.last_scores)                      # (L,H,C)
    assert torch.equal(o_scores[:, :, 1:], ref_scores), f"[{name}] retrieval scores differ"
    o_idx = torch.stack(ogc.last_idx)                            # (L,H,sets)
    ref_idx = torch.cat([torch.zeros_like(ref_topk_rest[:, :, :1]), ref_topk_rest + 1], dim=-1)
    for l in range(o_idx.shape[0]):
        assert R.topk_matches_reference(o_scores[l], o_idx[l], ref_idx[l]), f"[{name}] top-k layer {l}"
    same_order = torch.equal(o_idx, ref_idx)
    if same_order:
        assert torch.equal(ogc.key_cache, ref_retr_k) and torch.equal(ogc.value_cache, ref_retr_v)
    if temperature == 1.0 and top_p < 1e-6:
        # known-answer invariant: greedy TriForce == greedy autoregressive
        n = min(len(ar_tokens), len(streams[0]["tokens"]))
        assert streams[0]["tokens"][:n] == ar_tokens[:n], f"[{name}] greedy invariant violated"

    out = dict(
        name=name, tcfg=tcfg, dcfg=dcfg, tseed=tseed, dseed=dseed, pseed=pseed, head_std=head_std,
        prefill=prefill, budget=budget, gamma=gamma, chunk=chunk, gen_len=gen_len,
        temperature=temperature, top_p=top_p, rng_seed=rng_seed, repeats=repeats,
        ar_tokens=ar_tokens, triforce=streams, counts=[s["counts"] for s in o_streams],
        retrieval_scores=ref_scores, retrieval_idx=ref_idx, topk_order_identical=same_order,
        retr_k_digest=ref_retr_k.float().sum(dim=(1, 3)), retr_v_digest=ref_retr_v.float().sum(dim=(1, 3)),
        first_verify_logits=trace[0]["logits"] if trace else None,
        first_verify_tokens=trace[0]["verify_tokens"] if trace else None,
    )
    torch.save(out, os.path.join(GOLDEN, f"{name}.pt"))
    print(f"[{name}] ok: ref+oracle {t_ref:.1f}s, {len(ar_tokens)} AR tokens, acc "
          f"{[round(s['acceptance_rate'], 3) for s in streams]}, topk order identical: {same_order}")


def forward_case(name="forward_small"):
    """One target prefill+decode step and one draft spec step: logits must be bit-identical."""
    ref = _refshim.load_reference()
    tcfg = specs.tiny_target_config(vocab_size=512, layers=2, hidden=256, heads=2, max_pos=1024)
    dcfg = specs.llama_config(128, 256, 2, 2, vocab_size=512, max_position_embeddings=512, name="tiny-draft")
    tsd, dsd = specs.random_state_dict(tcfg, 11), specs.random_state_dict(dcfg, 12)
    prompt = specs.random_prompt(512, 200, 13)
    target = build_reference_model(ref, tcfg, tsd)
    draft = build_reference_model(ref, dcfg, dsd, draft=True)
    cache = ref.cache.FlashSimpleCache(target, 256)
    with torch.inference_mode():
        l1 = target(input_ids=prompt[:, :128], kv_cache=cache, graph_cache=None).logits
        l2 = target(input_ids=prompt[:, 128:200], kv_cache=cache, graph_cache=None).logits
        l3 = target(input_ids=prompt[:, :5], kv_cache=cache, graph_cache=None).logits
    ot = M.OracleTarget(tcfg, tsd)
    okv = M.FullCache(tcfg, 256)
    o1 = ot.forward(prompt[:, :128], okv)
    o2 = ot.forward(prompt[:, 128:200], okv)
    o3 = ot.forward(prompt[:, :5], okv)
    assert torch.equal(l1, o1) and torch.equal(l2, o2) and torch.equal(l3, o3), "target logits differ"
    dcache = ref.cache.StreamingLLMEvictionCache(draft, start_size=16, recent_size=100, gamma=4)
    odc = M.StreamingCacheO(dcfg, gamma=4, start_size=16, recent_size=100)
    od = M.OracleDraft(dcfg, dsd)
    with torch.inference_mode():
        for i in range(4):
            dcache.evict_prefill(50)
            dl = draft(input_ids=prompt[:, i * 50:(i + 1) * 50], kv_cache=dcache, graph_cache=None).logits
            odc.evict_prefill(50)
            ol = od.forward(prompt[:, i * 50:(i + 1) * 50], odc, None)
            assert torch.equal(dl, ol), f"draft prefill logits differ at block {i}"
        ds = draft(input_ids=prompt[:, :3], kv_cache=dcache, graph_cache=dcache, gamma_offset=2).logits
    os_ = od.forward(prompt[:, :3], odc, odc, gamma_offset=2)
    assert torch.equal(ds, os_), "draft spec logits differ"
    # sampling helpers
    g = torch.Generator().manual_seed(5)
    lg = torch.randn(6, 512, generator=g) * 3
    cases = [(0.6, 0.9), (1.0, 1e-9), (1.0, 1.0), (0.3, 0.5)]
    sm = {}
    for T, P in cases:
        a = ref.sampling.norm_logits(lg.clone(), temperature=T, top_k=-1, top_p=P)
        b = R.norm_logits(lg.clone(), temperature=T, top_k=-1, top_p=P)
        assert torch.equal(a, b), f"norm_logits({T},{P}) differs"
        sm[(T, P)] = a
    # exact ties: torch.sort(descending=True) is not stable (sampling.py:20), so WHICH of the tied
    # entries survives top-p is implementation-defined; the oracle's canonical rule is the stable
    # order (lowest index first).  Pin the tie-independent part: the sorted probability values.
    lt = lg.clone()
    lt[3, :7] = lt[3].max()
    lt[2] = lt[2].half().float()
    for T, P in cases:
        a = ref.sampling.norm_logits(lt.clone(), temperature=T, top_k=-1, top_p=P)
        b = R.norm_logits(lt.clone(), temperature=T, top_k=-1, top_p=P)
        assert torch.equal(torch.sort(a, dim=-1)[0], torch.sort(b, dim=-1)[0]), f"tie case ({T},{P})"
    x = torch.randn(512, generator=g)
    assert torch.equal(ref.sampling.max_fn(x), R.max_fn(x))
    torch.save(dict(tcfg=tcfg, dcfg=dcfg, tseed=11, dseed=12, pseed=13,
                    target_logits=[l1[:, -1].clone(), l2[:, -1].clone(), l3.clone()],
                    draft_spec_logits=ds.clone(), draft_prefill_last=dl[:, -1].clone(),
                    sampling_logits=lg, sampling_probs=sm, maxfn_in=x, maxfn_out=R.max_fn(x)),
               os.path.join(GOLDEN, f"{name}.pt"))
    print(f"[{name}] ok")


def rope_case(name="rope_tables"):
    ref = _refshim.load_reference()
    y = ref.modeling_llama.LlamaYaRNRotaryEmbedding(dim=128, max_position_embeddings=131072, base=10000,
                                                    scaling_factor=32.0, original_max_position_embeddings=4096)
    cos, sin = R.rope_tables_yarn(128, 131072, 32.0, 4096)
    assert torch.equal(y.cos_cached, cos) and torch.equal(y.sin_cached, sin), "yarn tables differ"
    p = ref.modeling_llama.LlamaRotaryEmbedding(128, max_position_embeddings=131072, base=1e7)
    pc, ps = R.rope_tables_plain(128, 131072, 1e7)
    assert torch.equal(p.cos_cached.half(), pc) and torch.equal(p.sin_cached.half(), ps), "plain tables differ"
    rows = torch.tensor([0, 1, 7, 4095, 4096, 65535, 124927, 131071])
    torch.save(dict(rows=rows, yarn_cos=cos[rows], yarn_sin=sin[rows], plain_cos=pc[rows], plain_sin=ps[rows],
                    yarn_mscale=float(y.mscale)), os.path.join(GOLDEN, f"{name}.pt"))
    print(f"[{name}] ok (mscale {y.mscale:.5f})")


def sequoia_case(name, tcfg, tseed, pseed, prefill, budget, chunk, gen_len, temperature, top_p, rng_seed,
                 branches=None, head_std=0.05):
    """Sequoia tree path: the UNMODIFIED reference SpecTree + TP_llama_tree.DistributedLlama + tensor_op run on
    CPU (1-rank gloo, torch proxies for the CUDA-only calls) vs the restatement in oracle/ref_tree.py."""
    import tempfile
    from oracle import ref_tree as RT
    ref = _refshim.load_reference_tree()
    tsd = specs.random_state_dict(tcfg, tseed, head_std=head_std)
    prompt = specs.random_prompt(tcfg["vocab_size"], prefill, pseed)[0]
    V = tcfg["vocab_size"]
    if branches is None:
        grow_map = torch.load(os.path.join(_refshim.REFERENCE_ROOT, "tree", "512.pt"))
        branches = grow_map["branches"]
        rebuilt = RT.grow_map_from_branches(branches)
        for k in ("roots", "branches", "Successors", "size"):
            assert rebuilt[k] == grow_map[k], f"grow_map_from_branches: {k} differs from tree/512.pt"
        assert torch.equal(rebuilt["mask"], grow_map["mask"]) and torch.equal(rebuilt["depth"], grow_map["depth"])
    else:
        grow_map = RT.grow_map_from_branches(branches)
    tree_size = grow_map["size"]
    L = tcfg["num_hidden_layers"]

    # ---------------- reference ----------------
    hf = build_reference_model(ref, tcfg, tsd)
    tmp = tempfile.mkdtemp()
    hf.config.save_pretrained(tmp)
    llm = ref.tree.DistributedLlama(model_name_or_path=tmp, local_rank=0, world_size=1, prefill=prefill, gen_len=gen_len,
                                    temperature=temperature, top_p=top_p, flash_attn=True, retrieval_budget=budget,
                                    retrieval_chunk_size=chunk, kv_offload=True, on_chip_layers=L - 1,
                                    tree_size=tree_size)
    llm.init_parameters(hf_model=hf)

    def get_residual(p, q):                               # test/offloading_seqouia.py:24-27
        r = (p - q).relu_()
        return r / (r.sum(dim=-1).unsqueeze(-1))

    def make_sampler(k):                                  # test/offloading_seqouia.py:29-39 (rank-0 branch)
        return lambda lg, rnd: (rnd.log() / torch.softmax(lg / temperature, dim=-1)).topk(k=k).indices.flatten()

    draft_step = len(grow_map["roots"])
    samplers = {i: make_sampler(max(grow_map["branches"][i])) for i in range(draft_step - 1)}
    gathers = {}
    for i in range(draft_step - 1):                       # :124-134
        mx = max(grow_map["branches"][i])
        gathers[i] = torch.cat([torch.arange(b) + j * mx for j, b in enumerate(grow_map["branches"][i])])
    torch.manual_seed(rng_seed)
    t0 = time.time()
    st = ref.spectree.SpecTree(engine=llm, temperature=temperature, top_p=top_p, max_length=prefill + gen_len,
                               grow_map=grow_map, residual_graph=get_residual, sampling_callables=samplers,
                               sample_gather_indices=gathers, tokenizer=_refshim.FakeTokenizer(), vocab_size=V)
    steps = []
    with torch.inference_mode():
        next_token = st.prefill(prefix=prompt)
        first = int(next_token)
        generated, n = [first], 0
        rand_table = st.rand.clone()
        while n < gen_len:                                # test/offloading_seqouia.py:155-185
            st.construct_grow_map(next_token=next_token)
            tree_tokens = st.verify_tokens.clone()
            draft0 = st.draft_logits[:8].clone()
            next_token, acc, toks = st.verify()
            if next_token is None:
                steps.append(dict(tree_tokens=tree_tokens, acc_count=acc, terminal=True))
                break
            generated.extend(toks[1:].tolist())
            steps.append(dict(tree_tokens=tree_tokens, acc_count=acc, accept_tokens=toks.tolist(), terminal=False,
                              draft_logits_head=draft0, seq_len=llm.kv_cache.seq_len))
            next_token = next_token.unsqueeze(0)
            n += acc
    t_ref = time.time() - t0

    # ---------------- oracle restatement ----------------
    eng = RT.TreeEngine(tcfg, tsd, prefill, gen_len, budget, chunk, tree_size)
    torch.manual_seed(rng_seed)
    so = RT.SpecTreeO(eng, grow_map, temperature, top_p, V, M.TorchRng(), rand=None)
    o_gen, o_counts = RT.run_sequoia(so, prompt, gen_len)
    assert o_gen == generated, f"[{name}] Sequoia stream mismatch\n{o_gen}\n{generated}"
    assert o_counts == [s["acc_count"] for s in steps if not s["terminal"]], f"[{name}] accept counts differ"
    for tr, s in zip(so.trace, steps):
        assert torch.equal(tr["tokens"], s["tree_tokens"]), f"[{name}] tree tokens differ"
    assert eng.kv_cache.seq_len == llm.kv_cache.seq_len
    torch.save(dict(name=name, tcfg=tcfg, tseed=tseed, pseed=pseed, head_std=head_std, prefill=prefill, budget=budget,
                    chunk=chunk, gen_len=gen_len, temperature=temperature, top_p=top_p, rng_seed=rng_seed,
                    branches=grow_map["branches"], tree_size=tree_size, mask_rowsum=grow_map["mask"].sum(dim=1),
                    depth=grow_map["depth"], generated=generated, steps=steps, rand_table_digest=rand_table.float().sum(dim=1),
                    final_seq_len=llm.kv_cache.seq_len), os.path.join(GOLDEN, f"{name}.pt"))
    print(f"[{name}] ok: reference {t_ref:.1f}s, {len(generated)} tokens in {len(steps)} steps, "
          f"accepts {[s['acc_count'] for s in steps]}")


def offloading_case(name="offloading_small"):
    """Single-GPU offloading entry (reference test/offloading.py:85-90): the reference's on-chip
    TriForce loop over OffloadingFlashSimpleCache (cache.py:63-115; KV in pinned host memory, one on-chip layer buffer,
    capacity prefill + gen_len + 32), constructed on CPU through the same torch proxy as the TP modules.  Known answer
    recorded with the stream: offloading changes where the KV lives, never the tokens — the stream must equal the
    FlashSimpleCache stream of the same seed (stochastic sampling, so every accept test and resample is compared)."""
    import json
    ref = _refshim.load_reference_tree()                    # installs the torch proxy in models.cache
    tcfg = specs.llama_config(256, 512, 3, 4, vocab_size=1024, max_position_embeddings=4096,
                              rope_scaling=dict(type="yarn", factor=8.0, original_max_position_embeddings=512),
                              name="tiny-d64-offload")
    dcfg = specs.llama_config(128, 256, 2, 2, vocab_size=1024, max_position_embeddings=2048, name="tiny-draft-offload")
    tseed, dseed, pseed, head_std = 501, 502, 503, 0.05
    prefill, budget, chunk, gamma, gen_len, temperature, top_p, rng_seed = 1000, 128, 8, 6, 32, 0.6, 0.9, 31
    tsd = specs.random_state_dict(tcfg, tseed, head_std=head_std)
    dsd = specs.random_state_dict(dcfg, dseed, head_std=head_std)
    prompt = specs.random_prompt(tcfg["vocab_size"], prefill, pseed)
    target = build_reference_model(ref, tcfg, tsd)
    draft = build_reference_model(ref, dcfg, dsd, draft=True)
    tok = _refshim.FakeTokenizer()
    tok.eos_token_id = -1
    emitted = []
    real = ref.decoding.spec_stream
    ref.decoding.spec_stream = lambda t, tk, color="blue": emitted.append(int(torch.as_tensor(t).reshape(-1)[0]))

    def run(cache):
        gcache = ref.cache.RetrievalCache(target, max_budget=budget, prefill=prefill, gamma=gamma, chunk_size=chunk)
        dcache = ref.cache.StreamingLLMEvictionCache(draft, start_size=16, recent_size=256 - 16 - gamma, gamma=gamma)
        eng = _refshim.EagerEngine(ref, target, cache, gcache, draft, dcache, temperature, top_p)
        emitted.clear()
        torch.manual_seed(rng_seed)
        acc, _ = ref.decoding.TriForce(tok, eng, prompt, gamma=gamma, max_len=gen_len, top_k=-1, top_p=top_p,
                                       temperature=temperature, verbose=True)
        return list(emitted), acc, int(cache.seq_len)

    try:
        off = run(ref.cache.OffloadingFlashSimpleCache(target, prefill + gen_len + 32))     # offloading.py:85
        res = run(ref.cache.FlashSimpleCache(target, prefill + gen_len + 16))               # on_chip.py:78
    finally:
        ref.decoding.spec_stream = real
    assert off == res, f"[{name}] offloading stream differs from the resident one\n{off}\n{res}"
    # oracle restatement (FullCache is the data model of both)
    oeng = M.OracleEngine(M.OracleTarget(tcfg, tsd), M.FullCache(tcfg, prefill + gen_len + 32),
                          M.RetrievalCacheO(tcfg, budget, prefill, chunk, gamma), M.OracleDraft(dcfg, dsd),
                          M.StreamingCacheO(dcfg, gamma=gamma, start_size=16, recent_size=256 - 16 - gamma),
                          temperature, top_p)
    torch.manual_seed(rng_seed)
    o = M.triforce(oeng, prompt, gamma, gen_len, temperature, top_p, eos_token_id=-1)
    assert o["tokens"] == off[0] and abs(o["acceptance_rate"] - off[1]) < 1e-12 and oeng.kv_cache.seq_len == off[2]
    with open(os.path.join(GOLDEN, name + ".json"), "w") as f:
        json.dump(dict(name=name, tcfg=tcfg, dcfg=dcfg, tseed=tseed, dseed=dseed, pseed=pseed, head_std=head_std,
                       prefill=prefill, budget=budget, chunk=chunk, gamma=gamma, gen_len=gen_len,
                       temperature=temperature, top_p=top_p, rng_seed=rng_seed, tokens=off[0], acceptance_rate=off[1],
                       final_seq_len=off[2], equals_resident_stream=True), f, indent=1)
    print(f"[{name}] ok: {len(off[0])} tokens, acceptance {off[1]:.3f}; offloading == resident stream")


def tp_chain_case(name="tp_chain", gamma=6, with_baselines=True):
    """Tensor-parallel chain path (SURVEY 8 rows a10/a11 `_Dist`, a13): the UNMODIFIED reference TP_llama.
    DistributedLlama + utils/decoding.TriForce_Dist / Middle_Spec_Dist run on CPU (1-rank gloo, torch proxies for the
    CUDA-only calls, one offloaded layer so the copy_kv / copy_back pipeline runs) vs ref_model.triforce(dist=True).
    Sub-cases: stochastic target, greedy target (the draft still samples at 0.6 / 0.9 — the reference's call sites
    never forward temperature / top_p), and two eos placements (:357-360 accepted eos, :382-383 loop exit)."""
    import tempfile
    ref = _refshim.load_reference_tp()
    tcfg = specs.llama_config(256, 512, 3, 4, vocab_size=1024, max_position_embeddings=4096,
                              rope_scaling=dict(type="yarn", factor=8.0, original_max_position_embeddings=512),
                              name="tiny-d64-tp")
    dcfg = specs.llama_config(128, 256, 2, 2, vocab_size=1024, max_position_embeddings=2048, name="tiny-draft-tp")
    tseed, dseed, pseed, head_std = 401, 402, 403, 0.05
    prefill, budget, chunk, gen_len = 1000, 128, 8, 40
    tsd = specs.random_state_dict(tcfg, tseed, head_std=head_std)
    dsd = specs.random_state_dict(dcfg, dseed, head_std=head_std)
    prompt = specs.random_prompt(tcfg["vocab_size"], prefill, pseed)
    hf = build_reference_model(ref, tcfg, tsd)
    draft = build_reference_model(ref, dcfg, dsd, draft=True)
    tmp = tempfile.mkdtemp()
    hf.config.save_pretrained(tmp)

    def reference_run(temperature, top_p, rng_seed, eos):
        dcache = ref.cache.StreamingLLMEvictionCache(draft, start_size=16, recent_size=256 - 16 - gamma, gamma=gamma)
        llm = ref.tp.DistributedLlama(model_name_or_path=tmp, local_rank=0, world_size=1, prefill=prefill,
                                      gen_len=gen_len, temperature=temperature, top_p=top_p, flash_attn=True,
                                      retrieval_budget=budget, retrieval_chunk_size=chunk, kv_offload=True,
                                      on_chip_layers=tcfg["num_hidden_layers"] - 1, draft=draft, draft_cache=dcache,
                                      gamma=gamma)
        llm.init_parameters(hf_model=hf)
        events = []
        real = ref.decoding.spec_stream
        ref.decoding.spec_stream = lambda t, tok, color="blue": events.append((color, int(torch.as_tensor(t).reshape(-1)[0])))
        tok = _refshim.FakeTokenizer()
        tok.eos_token_id = eos
        torch.manual_seed(rng_seed)
        try:
            avg_tokens, _ = ref.decoding.TriForce_Dist(tok, llm, prompt, gamma=gamma, max_len=gen_len, top_k=-1,
                                                       top_p=top_p, temperature=temperature, verbose=True)
        finally:
            ref.decoding.spec_stream = real
        assert events[0][0] == "cyan"
        tokens, counts, cur = [events[0][1]], [], 0
        for color, t in events[1:]:
            tokens.append(t)
            if color == "green":
                cur += 1
            else:                                   # red: rejection closes the step; blue: bonus token (+1, :392)
                counts.append(cur + (1 if color == "blue" else 0))
                cur = 0
        return dict(tokens=tokens, counts=counts, open_greens=cur, avg_tokens=avg_tokens,
                    final_seq_len=int(llm.kv_cache.seq_len), draft_seq_len=int(dcache.seq_len))

    def oracle_run(temperature, top_p, rng_seed, eos):
        eng = M.OracleEngine(M.OracleTarget(tcfg, tsd), M.FullCache(tcfg, prefill + gen_len + 32),
                             M.RetrievalCacheO(tcfg, budget, prefill, chunk, gamma), M.OracleDraft(dcfg, dsd),
                             M.StreamingCacheO(dcfg, gamma=gamma, start_size=16, recent_size=256 - 16 - gamma),
                             temperature, top_p, draft_chunk=128, draft_temperature=0.6, draft_top_p=0.9)
        torch.manual_seed(rng_seed)
        res = M.triforce(eng, prompt, gamma, gen_len, temperature, top_p, eos_token_id=eos, dist=True)
        return res, eng

    cases = []
    t0 = time.time()
    for label, temperature, top_p, rng_seed, eos_pick in (("stochastic", 0.6, 0.9, 11, None),
                                                          ("greedy_target", 1.0, 1e-9, 12, None),
                                                          ("eos_early", 0.6, 0.9, 11, 5),
                                                          ("eos_late", 0.6, 0.9, 11, 17)):
        eos = -1
        if eos_pick is not None:                    # a token that really occurs in the no-eos stream of this seed
            eos = cases[0]["tokens"][eos_pick]
        want = reference_run(temperature, top_p, rng_seed, eos)
        got, eng = oracle_run(temperature, top_p, rng_seed, eos)
        assert got["tokens"] == want["tokens"], f"[{name}/{label}] stream mismatch\n{got['tokens']}\n{want['tokens']}"
        ended_by_eos = eos in want["tokens"][1:]
        assert got["counts"][:len(want["counts"])] == want["counts"], f"[{name}/{label}] accept counts differ"
        assert abs(got["avg_tokens"] - want["avg_tokens"]) < 1e-12, f"[{name}/{label}] avg_tokens differ"
        if eos not in want["tokens"][1:]:
            assert eng.kv_cache.seq_len == want["final_seq_len"] and eng.draft_cache.seq_len == want["draft_seq_len"]
        cases.append(dict(label=label, temperature=temperature, top_p=top_p, rng_seed=rng_seed, eos=eos,
                          ended_by_eos=bool(ended_by_eos), **want))
        print(f"[{name}/{label}] ok: {len(want['tokens'])} tokens, counts {want['counts']}, avg {want['avg_tokens']:.4f}, "
              f"eos {eos}")
    # Baseline_Dist (decoding.py:243-287): the TP autoregressive baseline; torch.cuda.synchronize() is proxied for the
    # duration of the call only
    baselines = []
    for label, temperature, top_p, rng_seed in ((("baseline_stochastic", 0.6, 0.9, 13), ("baseline_greedy", 1.0, 1e-9, 14))
                                                 if with_baselines else ()):
        llm = ref.tp.DistributedLlama(model_name_or_path=tmp, local_rank=0, world_size=1, prefill=prefill,
                                      gen_len=gen_len, temperature=temperature, top_p=top_p, flash_attn=True,
                                      retrieval_budget=0, kv_offload=True,
                                      on_chip_layers=tcfg["num_hidden_layers"] - 1)      # offloading_TP.py:76
        llm.init_parameters(hf_model=hf)
        real_torch = ref.decoding.torch
        ref.decoding.torch = _refshim._TorchProxy()
        torch.manual_seed(rng_seed)
        try:
            _, gen_tokens = ref.decoding.Baseline_Dist(_refshim.FakeTokenizer(), llm, prompt, max_len=20,
                                                       temperature=temperature, top_p=top_p, local_rank=0)
        finally:
            ref.decoding.torch = real_torch
        eng = M.OracleEngine(M.OracleTarget(tcfg, tsd), M.FullCache(tcfg, prefill + gen_len + 32), None, None, None,
                             temperature, top_p)
        torch.manual_seed(rng_seed)
        o = M.autoregressive(eng, prompt, 20, temperature, top_p)
        assert o[1:] == gen_tokens[0].tolist(), f"[{name}/{label}] baseline stream mismatch"
        baselines.append(dict(label=label, temperature=temperature, top_p=top_p, rng_seed=rng_seed,
                              gen_tokens=gen_tokens[0].tolist(), final_seq_len=int(llm.kv_cache.seq_len)))
        print(f"[{name}/{label}] ok: {gen_tokens[0].tolist()[:8]}...")
    torch.save(dict(name=name, tcfg=tcfg, dcfg=dcfg, tseed=tseed, dseed=dseed, pseed=pseed, head_std=head_std,
                    prefill=prefill, budget=budget, chunk=chunk, gamma=gamma, gen_len=gen_len, cases=cases,
                    baselines=baselines),
               os.path.join(GOLDEN, f"{name}.pt"))
    print(f"[{name}] saved ({time.time() - t0:.1f}s)")


# ---- tensor-parallel engine at world size 2 (reference run as two gloo processes on CPU) -------------------------
_TP2 = dict(tcfg=dict(hidden=256, inter=512, layers=3, heads=4), tseed=601, dseed=602, pseed=603, head_std=0.05,
            prefill=1000, budget=128, chunk=8, gamma=6, gen_len=24, temperature=0.6, top_p=0.9, rng_seed=41)


# world size 8 needs 8 heads: hidden 512 = 8 x 64, 1024 MLP columns = 8 x 128 (one head and 8 column panels per rank)
_TP8 = dict(_TP2, tcfg=dict(hidden=512, inter=1024, layers=3, heads=8), tseed=611, dseed=612, pseed=613, rng_seed=43)
_TP_PARAMS = {"tp2": _TP2, "tp8": _TP8}


def _tp2_configs(which="tp2"):
    c = _TP_PARAMS[which]["tcfg"]
    tcfg = specs.llama_config(c["hidden"], c["inter"], c["layers"], c["heads"], vocab_size=1024,
                              max_position_embeddings=4096,
                              rope_scaling=dict(type="yarn", factor=8.0, original_max_position_embeddings=512),
                              name=f"tiny-d64-{which}")
    dcfg = specs.llama_config(128, 256, 2, 2, vocab_size=1024, max_position_embeddings=2048, name="tiny-draft-tp2")
    return tcfg, dcfg


def _tp2_worker(rank, world, port, out_path, which="tp2"):
    """One rank of the reference's TP engine (TP_layers.py:126-147 sharding, tensor_op.py all-reduces) on CPU."""
    import tempfile
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(4 if world <= 4 else 1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ref = _refshim.load_reference_tp()
    P = _TP_PARAMS[which]
    tcfg, dcfg = _tp2_configs(which)
    tsd = specs.random_state_dict(tcfg, P["tseed"], head_std=P["head_std"])
    dsd = specs.random_state_dict(dcfg, P["dseed"], head_std=P["head_std"])
    prompt = specs.random_prompt(tcfg["vocab_size"], P["prefill"], P["pseed"])
    gamma = P["gamma"]
    hf = build_reference_model(ref, tcfg, tsd)
    draft = build_reference_model(ref, dcfg, dsd, draft=True)
    dcache = ref.cache.StreamingLLMEvictionCache(draft, start_size=16, recent_size=256 - 16 - gamma, gamma=gamma)
    tmp = tempfile.mkdtemp()
    hf.config.save_pretrained(tmp)
    llm = ref.tp.DistributedLlama(model_name_or_path=tmp, local_rank=rank, world_size=world, prefill=P["prefill"],
                                  gen_len=P["gen_len"], temperature=P["temperature"], top_p=P["top_p"], flash_attn=True,
                                  retrieval_budget=P["budget"], retrieval_chunk_size=P["chunk"], kv_offload=True,
                                  on_chip_layers=tcfg["num_hidden_layers"] - 1, draft=draft, draft_cache=dcache,
                                  gamma=gamma)
    llm.init_parameters(hf_model=hf)
    shard_shapes = {k: tuple(getattr(llm.layers[0], k).shape) for k in ("wq", "wk", "wv", "wo", "gate_proj", "up_proj",
                                                                         "down_proj")}
    llm.reset()
    prefill_logits = llm.prefill(prompt[:, :-1])[:, -1].clone()
    # the chunks THIS rank's heads select (cache.py:531-536: torch.topk over the local heads' chunk scores), layer by layer:
    # a near-tie at the k-th score may legitimately resolve differently on another implementation's scores, and one swapped
    # chunk moves the retrieval-verify logits by more than any rounding tolerance — the product's tests compare their own
    # selection tie-tolerantly and then run the verify stage over the reference's selection
    topk_log, real_topk = [], torch.topk

    def spy_topk(x, k, dim=-1, **kw):
        out = real_topk(x, k=k, dim=dim, **kw)
        if x.dim() == 3 and x.dtype == torch.float16:
            topk_log.append(torch.cat([torch.zeros_like(out[1][0, :, :1]), out[1][0] + 1], dim=-1).clone())
        return out
    torch.topk = spy_topk
    try:
        build_logits = llm.build_retrieval_cache(prompt[:, -1:]).clone()
    finally:
        torch.topk = real_topk
    assert len(topk_log) == tcfg["num_hidden_layers"], len(topk_log)
    S = int(llm.kv_cache.seq_len)
    vt = torch.tensor([[11, 12, 13] + [100] * (gamma - 2)])
    pos = torch.arange(S, S + gamma + 1).unsqueeze(0)
    spec_logits = llm.retrieval_inference(vt, pos).clone()
    verify_logits = llm.inference(vt).clone()
    llm.kv_cache.seq_len -= gamma + 1
    events = []
    ref.decoding.spec_stream = lambda t, tok, color="blue": events.append((color, int(torch.as_tensor(t).reshape(-1)[0])))
    tok = _refshim.FakeTokenizer()
    tok.eos_token_id = -1
    torch.manual_seed(P["rng_seed"])
    avg, _ = ref.decoding.TriForce_Dist(tok, llm, prompt, gamma=gamma, max_len=P["gen_len"], top_k=-1, top_p=P["top_p"],
                                        temperature=P["temperature"], verbose=True)
    tokens, counts, cur = [events[0][1]], [], 0
    for color, t in events[1:]:
        tokens.append(t)
        if color == "green":
            cur += 1
        else:
            counts.append(cur + (1 if color == "blue" else 0))
            cur = 0
    torch.save(dict(rank=rank, shard_shapes=shard_shapes, topk_idx=topk_log, prefill_logits=prefill_logits, build_logits=build_logits,
                    spec_logits=spec_logits, verify_logits=verify_logits, S=S, tokens=tokens, counts=counts,
                    avg_tokens=avg, final_seq_len=int(llm.kv_cache.seq_len)), f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def tp_world2_case(name="tp_world2", world=2, which="tp2"):
    """The reference's tensor-parallel engine at WORLD SIZE 2: two gloo processes on CPU run the unmodified
    TP_llama.DistributedLlama (head / MLP-column shards of TP_layers.py:126-147, fp16 all-reduce after wo and down_proj)
    and TriForce_Dist (rank 0 samples, tokens and uniforms broadcast).  Recorded: per-stage logits (identical on both
    ranks), the shard shapes, the token stream.  Checked here: both ranks agree bit-for-bit; the single-process
    restatement is within fp16 all-reduce rounding of the logits and reproduces the stream."""
    import socket
    import tempfile
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = os.path.join(tempfile.mkdtemp(), "tp2")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_tp2_worker, args=(r, world, port, out, which)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=1800)
        assert p.exitcode == 0, f"reference TP rank exited with {p.exitcode}"
    ranks = [torch.load(f"{out}.{r}", weights_only=False) for r in range(world)]
    r0 = ranks[0]
    for r1 in ranks[1:]:
        for k in ("prefill_logits", "build_logits", "spec_logits", "verify_logits"):
            assert torch.equal(r0[k], r1[k]), f"[{name}] ranks disagree on {k}"
        assert r0["tokens"] == r1["tokens"] and r0["counts"] == r1["counts"] and r0["final_seq_len"] == r1["final_seq_len"]
    P = _TP_PARAMS[which]
    tcfg, dcfg = _tp2_configs(which)
    g = dict(name=name, world=world, tcfg=tcfg, dcfg=dcfg, **{k: v for k, v in P.items() if k != "tcfg"})
    from tests import helpers as Hh
    eng, _, _ = Hh.build_oracle_tp(g, P["temperature"], P["top_p"])
    prompt = specs.random_prompt(tcfg["vocab_size"], P["prefill"], P["pseed"])
    gamma = P["gamma"]
    lp = eng.inference(prompt[:, :-1])[:, -1]
    lb = eng.inference(prompt[:, -1:])
    S = eng.kv_cache.seq_len
    vt = torch.tensor([[11, 12, 13] + [100] * (gamma - 2)])
    ls = eng.model.forward(vt, eng.kv_cache, eng.graph_cache, position_ids=torch.arange(S, S + gamma + 1).unsqueeze(0),
                           spec=True)
    lv = eng.inference(vt)
    gaps = {}
    for k, ours in (("prefill_logits", lp), ("build_logits", lb), ("spec_logits", ls), ("verify_logits", lv)):
        gaps[k] = float((ours - r0[k]).abs().max())
        # (an 8-way fp16 all-reduce rounds seven partial sums where one process rounds none: twice the allowance of world <= 4)
        assert gaps[k] < (4e-3 if world <= 4 else 8e-3), f"[{name}] single-process restatement off by {gaps[k]:.2e} on {k}"
    eng, _, _ = Hh.build_oracle_tp(g, P["temperature"], P["top_p"])
    torch.manual_seed(P["rng_seed"])
    res = M.triforce(eng, prompt, gamma, P["gen_len"], P["temperature"], P["top_p"], eos_token_id=-1, dist=True)
    same_stream = res["tokens"] == r0["tokens"]
    g.update(topk_idx=[r["topk_idx"] for r in ranks],           # [rank][layer] (H / world, select_sets), chunk 0 first
             shard_shapes=r0["shard_shapes"], prefill_logits=r0["prefill_logits"], build_logits=r0["build_logits"],
             spec_logits=r0["spec_logits"], verify_logits=r0["verify_logits"], S=r0["S"], tokens=r0["tokens"],
             counts=r0["counts"], avg_tokens=r0["avg_tokens"], final_seq_len=r0["final_seq_len"],
             single_process_gaps=gaps, single_process_stream_identical=same_stream,
             common_prefix=next((i for i, (a, b) in enumerate(zip(res["tokens"], r0["tokens"])) if a != b),
                                min(len(res["tokens"]), len(r0["tokens"]))))
    torch.save(g, os.path.join(GOLDEN, f"{name}.pt"))
    print(f"[{name}] ok: ranks bit-identical; single-process gaps {gaps}; stream identical: {same_stream} "
          f"(common prefix {g['common_prefix']} of {len(r0['tokens'])}); shards {r0['shard_shapes']}")


# ---- Sequoia tree path at world size 2 (reference run as two gloo processes on CPU) ------------------------------
_SEQ2 = dict(tseed=711, pseed=712, head_std=0.05, prefill=600 - 600 % 8, budget=64, chunk=8, gen_len=16,
             temperature=0.8, top_p=0.95, rng_seed=19,
             branches=[[4], [3, 2, 0, 1], [2, 1, 1, 1, 0, 1], [1, 1, 0, 1, 0, 0], [1, 0, 0]])


def _seq2_config():
    return specs.llama_config(256, 512, 3, 4, vocab_size=1024, max_position_embeddings=4096,
                              rope_scaling=dict(type="yarn", factor=8.0, original_max_position_embeddings=512),
                              name="tiny-d64-tree2")


def _seq2_worker(rank, world, port, out_path):
    """One rank of test/offloading_seqouia.py's loop (:155-185) on the reference's TP_llama_tree engine, CPU + gloo."""
    import tempfile
    import torch.distributed as dist
    from oracle import ref_tree as RT
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(4)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ref = _refshim.load_reference_tree()
    P = _SEQ2
    tcfg = _seq2_config()
    tsd = specs.random_state_dict(tcfg, P["tseed"], head_std=P["head_std"])
    prompt = specs.random_prompt(tcfg["vocab_size"], P["prefill"], P["pseed"])[0]
    grow_map = RT.grow_map_from_branches(P["branches"])
    temperature, V = P["temperature"], tcfg["vocab_size"]
    hf = build_reference_model(ref, tcfg, tsd)
    tmp = tempfile.mkdtemp()
    hf.config.save_pretrained(tmp)
    llm = ref.tree.DistributedLlama(model_name_or_path=tmp, local_rank=rank, world_size=world, prefill=P["prefill"],
                                    gen_len=P["gen_len"], temperature=temperature, top_p=P["top_p"], flash_attn=True,
                                    retrieval_budget=P["budget"], retrieval_chunk_size=P["chunk"], kv_offload=True,
                                    on_chip_layers=tcfg["num_hidden_layers"] - 1, tree_size=grow_map["size"])
    llm.init_parameters(hf_model=hf)

    def get_residual(p, q):                               # test/offloading_seqouia.py:24-27
        r = (p - q).relu_()
        return r / (r.sum(dim=-1).unsqueeze(-1))

    def make_sampler(k):                                  # test/offloading_seqouia.py:29-39, both branches
        def run(lg, rnd):
            if dist.get_rank() == 0:
                position = (rnd.log() / torch.softmax(lg / temperature, dim=-1)).topk(k=k).indices.flatten()
            else:
                position = torch.full((k * lg.shape[0],), -1, dtype=torch.long)
            dist.broadcast(position, src=0)
            return position
        return run

    draft_step = len(grow_map["roots"])
    samplers = {i: make_sampler(max(grow_map["branches"][i])) for i in range(draft_step - 1)}
    gathers = {i: torch.cat([torch.arange(b) + j * max(grow_map["branches"][i])
                             for j, b in enumerate(grow_map["branches"][i])]) for i in range(draft_step - 1)}
    torch.manual_seed(P["rng_seed"])
    st = ref.spectree.SpecTree(engine=llm, temperature=temperature, top_p=P["top_p"],
                               max_length=P["prefill"] + P["gen_len"], grow_map=grow_map, residual_graph=get_residual,
                               sampling_callables=samplers, sample_gather_indices=gathers,
                               tokenizer=_refshim.FakeTokenizer(), vocab_size=V)
    seen = []
    real_inference = llm.inference

    def spy_inference(*a, **k):
        out = real_inference(*a, **k)
        seen.append(out.clone())
        return out

    steps = []
    with torch.inference_mode():
        next_token = st.prefill(prefix=prompt)
        first = int(next_token)
        rand_table = st.rand.clone()
        generated, n = [first], 0
        while n < P["gen_len"]:
            st.construct_grow_map(next_token=next_token)
            rec = dict(tree_tokens=st.verify_tokens.clone(), seq_len=int(llm.kv_cache.seq_len))
            if not steps:
                rec["draft_logits"] = st.draft_logits.clone()
                llm.inference = spy_inference
            next_token, acc, toks = st.verify()
            if not steps:
                llm.inference = real_inference
                rec["verify_logits"] = seen[-1][0].clone()
            if next_token is None:
                rec.update(acc_count=acc, terminal=True)
                steps.append(rec)
                break
            generated.extend(toks[1:].tolist())
            rec.update(acc_count=acc, accept_tokens=toks.tolist(), terminal=False)
            steps.append(rec)
            next_token = next_token.unsqueeze(0)
            n += acc
    torch.save(dict(rank=rank, first=first, rand_table=rand_table, generated=generated, steps=steps,
                    final_seq_len=int(llm.kv_cache.seq_len)), f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def sequoia_world2_case(name="sequoia_world2"):
    """test/offloading_seqouia.py's loop at WORLD SIZE 2: the unmodified reference SpecTree + TP_llama_tree engine as two
    gloo processes on CPU (rank 0 draws the children and decides the walk, everything is broadcast).  Recorded: the
    uniform table of the tree growth, the first step's per-node draft logits and 20-row verify logits, every step's
    tree tokens and accept counts.  Checked here: ranks agree bit-for-bit; the single-process restatement reproduces
    the stream (reported, the all-reduce may flip a near-tie)."""
    import socket
    import tempfile
    import torch.multiprocessing as mp
    from oracle import ref_tree as RT
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = os.path.join(tempfile.mkdtemp(), "seq2")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_seq2_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=900)
        assert p.exitcode == 0, f"reference Sequoia rank exited with {p.exitcode}"
    r0, r1 = (torch.load(f"{out}.{r}", weights_only=False) for r in range(2))
    assert r0["generated"] == r1["generated"] and r0["final_seq_len"] == r1["final_seq_len"]
    for a, b in zip(r0["steps"], r1["steps"]):
        assert torch.equal(a["tree_tokens"], b["tree_tokens"]) and a["acc_count"] == b["acc_count"]
    assert torch.equal(r0["steps"][0]["draft_logits"], r1["steps"][0]["draft_logits"])
    assert torch.equal(r0["steps"][0]["verify_logits"], r1["steps"][0]["verify_logits"])
    P = _SEQ2
    tcfg = _seq2_config()
    grow_map = RT.grow_map_from_branches(P["branches"])
    tsd = specs.random_state_dict(tcfg, P["tseed"], head_std=P["head_std"])
    prompt = specs.random_prompt(tcfg["vocab_size"], P["prefill"], P["pseed"])[0]
    eng = RT.TreeEngine(tcfg, tsd, P["prefill"], P["gen_len"], P["budget"], P["chunk"], grow_map["size"])
    torch.manual_seed(P["rng_seed"])
    so = RT.SpecTreeO(eng, grow_map, P["temperature"], P["top_p"], tcfg["vocab_size"], M.TorchRng(), rand=None)
    o_gen, o_counts = RT.run_sequoia(so, prompt, P["gen_len"])
    same = o_gen == r0["generated"]
    cp = next((i for i, (a, b) in enumerate(zip(o_gen, r0["generated"])) if a != b), min(len(o_gen), len(r0["generated"])))
    g = dict(name=name, tcfg=tcfg, tree_size=grow_map["size"], single_process_stream_identical=same, common_prefix=cp,
             first=r0["first"], rand_table=r0["rand_table"], generated=r0["generated"], steps=r0["steps"],
             final_seq_len=r0["final_seq_len"], **P)
    torch.save(g, os.path.join(GOLDEN, f"{name}.pt"))
    print(f"[{name}] ok: ranks bit-identical, {len(r0['generated'])} tokens in {len(r0['steps'])} steps, accepts "
          f"{[s['acc_count'] for s in r0['steps']]}; single-process stream identical: {same} (common prefix {cp})")


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    torch.set_num_threads(8)
    rope_case()
    forward_case()
    # BASELINE.json configs[0]: Llama-68M-shaped draft + tiny target, prefill 2048, budget 256, gamma 4, greedy
    triforce_case("cfg1_greedy", specs.tiny_target_config(), specs.draft_68m_config(), 101, 102, 103,
                  prefill=2048, budget=256, gamma=4, chunk=8, gen_len=48, temperature=1.0, top_p=1e-9, repeats=2)
    # same plumbing, stochastic sampling (cfg3-style T=0.6/top_p=0.9), seeded torch RNG replayed exactly
    triforce_case("cfg1_stochastic", specs.tiny_target_config(), specs.draft_68m_config(), 101, 102, 103,
                  prefill=2048, budget=256, gamma=4, chunk=8, gen_len=32, temperature=0.6, top_p=0.9, rng_seed=77)
    # small-vocab, D=64 heads, gamma 6 (cfg2's gamma), ragged prompt length (not a multiple of 128)
    triforce_case("small_gamma6",
                  specs.llama_config(256, 512, 3, 4, vocab_size=1024, max_position_embeddings=4096,
                                     rope_scaling=dict(type="yarn", factor=8.0, original_max_position_embeddings=512),
                                     name="tiny-d64"),
                  specs.llama_config(128, 256, 2, 2, vocab_size=1024, max_position_embeddings=2048, name="tiny-draft"),
                  201, 202, 203, prefill=1000, budget=128, gamma=6, chunk=8, gen_len=40,
                  temperature=1.0, top_p=1e-9, repeats=1)


def shards_case(name="tp_shards"):
    """The reference's own weight slicing (models/TP_layers.py:126-147, DistributedLlamaLayer.init_parameters) for EVERY
    rank of a 4-way and an 8-way split of one seeded layer, next to oracle.specs.shard_of — the helper the 13B / TP = 8
    shard-width GPU parity test builds its oracle from.  Recorded: per (world, rank) the seven shard tensors of the
    reference as (shape, sha256 of the fp16 bytes) — the tensors are functions of (config, seed) and are rebuilt by the test.
    Checked here: shard_of returns exactly those tensors."""
    ref = _refshim.load_reference_tp()
    TPL = ref.tp_layers if hasattr(ref, "tp_layers") else __import__("models.TP_layers", fromlist=["x"])
    # 13B-like proportions, scaled down: 40 heads (D = 16), hidden 640, intermediate 1728 = 8 * 216
    cfg = specs.llama_config(640, 1728, 1, 40, vocab_size=256, max_position_embeddings=4096,
                             rope_scaling=dict(type="yarn", factor=8.0, original_max_position_embeddings=512),
                             name="tiny-13b-proportions")
    sd = specs.random_state_dict(cfg, 77)
    hf = build_reference_model(ref, cfg, sd)
    out = dict(cfg=cfg, seed=77, cases=[])
    for world in (4, 8):
        for rank in range(world):
            dcfg = TPL.DistributedOffloadingConfig(hf.config, local_rank=rank, world_size=world)
            layer = TPL.DistributedLlamaLayer(0, dcfg)
            layer.init_parameters(hf.model.layers[0])
            theirs = dict(q=layer.wq, k=layer.wk, v=layer.wv, o=layer.wo, gate=layer.gate_proj, up=layer.up_proj,
                          down=layer.down_proj)
            scfg, ssd = specs.shard_of(cfg, sd, rank, world)
            pre = "model.layers.0."
            ours = dict(q=ssd[pre + "self_attn.q_proj.weight"], k=ssd[pre + "self_attn.k_proj.weight"],
                        v=ssd[pre + "self_attn.v_proj.weight"], o=ssd[pre + "self_attn.o_proj.weight"],
                        gate=ssd[pre + "mlp.gate_proj.weight"], up=ssd[pre + "mlp.up_proj.weight"],
                        down=ssd[pre + "mlp.down_proj.weight"])
            for kname in theirs:
                assert torch.equal(theirs[kname].cpu(), ours[kname]), (world, rank, kname)
            assert scfg["num_attention_heads"] == 40 // world and scfg["intermediate_size"] == 1728 // world
            out["cases"].append(dict(world=world, rank=rank,
                                     shards={k: dict(shape=list(v.shape), sha256=specs.tensor_digest(v)) for k, v in theirs.items()}))
    import json
    with open(os.path.join(GOLDEN, f"{name}.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(f"[golden] {name}: {len(out['cases'])} (world, rank) shard sets written (digests); shard_of == reference slicing")


def cli_case(name="cli_flags"):
    """The command lines of the reference's four entry scripts (flag -> type name, default), read from their
    `add_argument` calls with `ast` (the scripts cannot be imported: they run their benchmark at import time)."""
    import ast
    import json
    out = {}
    for script in ("on_chip", "offloading", "offloading_TP", "offloading_seqouia"):
        tree = ast.parse(open(os.path.join(_refshim.REFERENCE_ROOT, "test", script + ".py")).read())
        flags = {}
        for node in ast.walk(tree):
            if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "add_argument":
                flag = node.args[0].value
                kw = {k.arg: k.value for k in node.keywords}
                if "action" in kw:
                    flags[flag] = ["flag", None]
                else:
                    typ = kw["type"].id if "type" in kw else "str"
                    flags[flag] = [typ, ast.literal_eval(kw["default"]) if "default" in kw else None]
        out[script] = flags
    with open(os.path.join(GOLDEN, name + ".json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(f"[{name}] ok: " + ", ".join(f"{k}: {len(v)} flags" for k, v in out.items()))


def main_sequoia():
    os.makedirs(GOLDEN, exist_ok=True)
    torch.set_num_threads(8)
    small = specs.llama_config(256, 512, 2, 2, vocab_size=1024, max_position_embeddings=4096,
                               rope_scaling=dict(type="yarn", factor=8.0, original_max_position_embeddings=512),
                               name="tiny-d128-tree")
    # the reference's own 512-node tree (tree/512.pt) on a tiny D=128 target
    sequoia_case("sequoia_tree512", small, 301, 302, prefill=1024, budget=128, chunk=8, gen_len=24,
                 temperature=0.6, top_p=0.9, rng_seed=5)
    # a small hand-made tree (ragged fan-out, a leaf on level 1) and D=64 heads
    d64 = specs.llama_config(256, 512, 3, 4, vocab_size=1024, max_position_embeddings=4096,
                             rope_scaling=dict(type="yarn", factor=8.0, original_max_position_embeddings=512),
                             name="tiny-d64-tree")
    sequoia_case("sequoia_small", d64, 311, 312, prefill=600 - 600 % 8, budget=64, chunk=8, gen_len=20,
                 temperature=0.8, top_p=0.95, rng_seed=9,
                 branches=[[4], [3, 2, 0, 1], [2, 1, 1, 1, 0, 1], [1, 1, 0, 1, 0, 0], [1, 0, 0]])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "sequoia":
        main_sequoia()
    elif len(sys.argv) > 1 and sys.argv[1] == "cli":
        os.makedirs(GOLDEN, exist_ok=True)
        cli_case()
    elif len(sys.argv) > 1 and sys.argv[1] == "tp":
        os.makedirs(GOLDEN, exist_ok=True)
        torch.set_num_threads(8)
        tp_chain_case()
        tp_chain_case("tp_chain_gamma16", gamma=16, with_baselines=False)     # offloading_TP.py's README command
    elif len(sys.argv) > 1 and sys.argv[1] == "shards":
        shards_case()
    elif len(sys.argv) > 1 and sys.argv[1] == "tp2":
        os.makedirs(GOLDEN, exist_ok=True)
        tp_world2_case()
        tp_world2_case("tp_world4", world=4)              # one attention head per rank: the extreme shard
    elif len(sys.argv) > 1 and sys.argv[1] == "tp8":
        os.makedirs(GOLDEN, exist_ok=True)
        tp_world2_case("tp_world8", world=8, which="tp8")  # BASELINE configs[4]'s world size (8 heads: one per rank)
    elif len(sys.argv) > 1 and sys.argv[1] == "sequoia2":
        os.makedirs(GOLDEN, exist_ok=True)
        sequoia_world2_case()
    elif len(sys.argv) > 1 and sys.argv[1] == "offloading":
        os.makedirs(GOLDEN, exist_ok=True)
        torch.set_num_threads(8)
        offloading_case()
    else:
        main()
        main_sequoia()
        cli_case()
        tp_chain_case()
        tp_chain_case("tp_chain_gamma16", gamma=16, with_baselines=False)
        offloading_case()
        tp_world2_case()
        tp_world2_case("tp_world4", world=4)
        tp_world2_case("tp_world8", world=8, which="tp8")
        sequoia_world2_case()
