"""TP engine + offloading tier on one MI355X (RCCL world_size 1): the pinned-host KV streaming path
(hipMemcpy2DAsync on the copy stream, event-ordered double buffering, D2H write-back of new tokens) must
give bit-identical logits to the same engine with every layer resident in HBM, and both must match the oracle."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from oracle import specs
from tests import helpers as Hh

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pg():
    if not dist.is_initialized():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=0, world_size=1, init_method=f"tcp://127.0.0.1:{port}")


def _build(g, on_chip, gamma=None, graphs=False):
    from triforce_amd.models.cache import StreamingLLMEvictionCache
    from triforce_amd.models.config_yarn import LlamaConfig
    from triforce_amd.models.modeling_llama_68m import LlamaForCausalLM as Draft
    from triforce_amd.models.TP_llama import DistributedLlama
    tsd = specs.random_state_dict(g["tcfg"], g["tseed"], head_std=g["head_std"])
    dsd = specs.random_state_dict(g["dcfg"], g["dseed"], head_std=g["head_std"])
    gamma = gamma or g["gamma"]
    draft = Draft.from_state_dict(LlamaConfig.from_dict(g["dcfg"]), dsd, DEV)
    dcache = StreamingLLMEvictionCache(draft, start_size=16, recent_size=256 - 16 - gamma, gamma=gamma)
    llm = DistributedLlama("unused", config=LlamaConfig.from_dict(g["tcfg"]), device=DEV, local_rank=0, world_size=1,
                           prefill=g["prefill"], gen_len=g["gen_len"], temperature=g["temperature"], top_p=g["top_p"],
                           retrieval_budget=g["budget"], kv_offload=True, on_chip_layers=on_chip, draft=draft,
                           draft_cache=dcache, gamma=gamma)
    llm.init_parameters(tsd)
    if graphs:
        llm.initialize_graphs()
    return llm


def _run(llm, g):
    from triforce_amd.utils.decoding import TriForce_Dist
    prompt = Hh.prompt_of(g).to(DEV)
    llm.reset()
    llm.prefill(prompt[:, :-1])
    logits = llm.build_retrieval_cache(prompt[:, -1:])
    S = llm.kv_cache.seq_len
    gamma = g["gamma"]
    vt = torch.tensor([[11, 12, 13] + [100] * (gamma - 2)], device=DEV)
    spec = llm.retrieval_inference(vt, torch.arange(S, S + gamma + 1, device=DEV).unsqueeze(0))
    step = llm.inference(vt)                                   # a verify-sized target forward (streams offloaded layers)
    res = TriForce_Dist(Hh.FakeTokenizer(), llm, prompt, gamma=gamma, max_len=20, top_k=-1, top_p=g["top_p"],
                        temperature=g["temperature"], return_details=True)
    torch.cuda.synchronize()
    return logits.cpu(), spec.cpu(), step.cpu(), res


def test_offloaded_layers_equal_resident_layers_and_oracle():
    _pg()
    g = Hh.load_golden("small_gamma6")
    L = g["tcfg"]["num_hidden_layers"]
    resident = _run(_build(g, on_chip=L), g)
    for on_chip in (0, 1, L - 1):
        off = _run(_build(g, on_chip=on_chip), g)
        assert torch.equal(off[0], resident[0]), f"on_chip={on_chip}: prefill logits differ from the resident run"
        assert torch.equal(off[1], resident[1]), f"on_chip={on_chip}: retrieval-verify logits differ"
        assert torch.equal(off[2], resident[2]), f"on_chip={on_chip}: target-verify logits differ"
        assert off[3]["tokens"] == resident[3]["tokens"] and off[3]["counts"] == resident[3]["counts"]
    gaps = Hh.teacher_forced_gaps(g, resident[3]["tokens"])
    assert max(gaps) < 8e-3
    assert Hh.common_prefix(resident[3]["tokens"], g["ar_tokens"]) >= 12
    # host copy of an offloaded layer holds exactly what the device computed (write-back path)
    llm0, llmL = _build(g, on_chip=0), _build(g, on_chip=L)
    p = Hh.prompt_of(g).to(DEV)
    for m in (llm0, llmL):
        m.reset()
        m.prefill(p[:, :300])
    torch.cuda.synchronize()
    assert torch.equal(llm0.kv_cache.cpu_k[:, :, :300], llmL.kv_cache.k[:, :, :300].cpu())
    assert torch.equal(llm0.kv_cache.cpu_v[:, :, :300], llmL.kv_cache.v[:, :, :300].cpu())


def test_gamma16_two_query_tiles():
    """cfg4/cfg5 use gamma=16: retrieval verify has 17 query rows and the target verify 17-18, i.e. the two-tile
    (32-row) instantiation of the attention kernel and the MT=2 skinny GEMM, inside the engine."""
    from triforce_amd.utils.decoding import TriForce_Dist
    _pg()
    g = Hh.load_golden("small_gamma6")
    L = g["tcfg"]["num_hidden_layers"]
    llm = _build(g, on_chip=L, gamma=16)
    prompt = Hh.prompt_of(g).to(DEV)
    res = TriForce_Dist(Hh.FakeTokenizer(), llm, prompt, gamma=16, max_len=24, top_k=-1, top_p=g["top_p"],
                        temperature=g["temperature"], return_details=True)
    gaps = Hh.teacher_forced_gaps(g, res["tokens"])
    assert max(gaps) < 8e-3, f"gamma=16 stream leaves the oracle's greedy path (gap {max(gaps):.4f})"
    assert Hh.common_prefix(res["tokens"], g["ar_tokens"]) >= 12


@pytest.mark.parametrize("form", ["whole", "segments"])
def test_tp_graphs_equal_eager(form, monkeypatch):
    """Captured draft steps + captured retrieval verify + captured target verify (device-side cache length), in the
    whole-forward form and in the collective-free segment form (the multi-rank fallback; with one rank the RCCL
    calls are no-ops) == the eager engine."""
    from tests.test_gpu_e2e import _logit_check
    _pg()
    monkeypatch.setenv("TRIFORCE_TP_GRAPHS", form)
    g = Hh.load_golden("small_gamma6")
    L = g["tcfg"]["num_hidden_layers"]
    graphed_llm = _build(g, on_chip=L, graphs=True)
    assert graphed_llm.graph_form == form and len(graphed_llm._draft_graphs) == g["gamma"] + 3
    assert sorted(graphed_llm._target_caps) == [g["gamma"] + 1, g["gamma"] + 2]
    if form == "segments":
        assert len(graphed_llm._verify_cap["graphs"]) == 2 * L + 1
    graphed = _run(graphed_llm, g)
    monkeypatch.setenv("TRIFORCE_TP_GRAPHS", "0")
    eager_llm = _build(g, on_chip=L, graphs=True)
    assert eager_llm.graph_form == "eager" and not eager_llm._target_caps
    eager = _run(eager_llm, g)
    assert torch.equal(eager[0], graphed[0]) and torch.equal(eager[1], graphed[1])
    # the captured target verify sizes its KV splits by the cache capacity, the eager one by the live length:
    # same math, different fp32 summation order
    _logit_check(f"{form}: captured target verify", graphed[2], eager[2])
    assert Hh.common_prefix(eager[3]["tokens"], graphed[3]["tokens"]) >= 12


def test_tp_auto_mode_single_rank_is_whole_graph(monkeypatch):
    _pg()
    monkeypatch.delenv("TRIFORCE_TP_GRAPHS", raising=False)
    g = Hh.load_golden("small_gamma6")
    llm = _build(g, on_chip=g["tcfg"]["num_hidden_layers"], graphs=True)
    assert llm.graph_form == "whole"
    # the probe used by the multi-rank auto mode: capture + one replay checked against the eager forward
    llm.reset()
    prompt = Hh.prompt_of(g).to(DEV)
    llm.prefill(prompt[:, :-1])
    llm.build_retrieval_cache(prompt[:, -1:])
    assert llm._try_whole(g["gamma"] + 1, "retrieval", False) is not None
    assert llm._try_whole(g["gamma"] + 1, "target", False) is not None


def test_single_gpu_offloading_cache_equals_resident_cache():
    """test/offloading.py path: OffloadingFlashSimpleCache (all KV in pinned host memory, streamed per forward)
    must reproduce the on-chip FlashSimpleCache run bit for bit — same kernels, the data only travels."""
    from triforce_amd.models.cache import OffloadingFlashSimpleCache
    from triforce_amd.utils.decoding import Autoregressive, TriForce
    g = Hh.load_golden("small_gamma6")
    prompt = Hh.prompt_of(g).to(DEV)
    tok = Hh.FakeTokenizer()
    # both engines run the target verify eagerly: the resident engine's captured form picks its split count from the
    # cache CAPACITY, not the live length — same math, another fp32 summation order; tested against the eager form in
    # test_gpu_e2e.test_captured_target_verify_equals_eager — and the bit-for-bit claim here is about where the data lives
    ge_res = Hh.build_product(g, DEV, graphs=True, target_graph=False)
    want = TriForce(tok, ge_res, prompt, gamma=g["gamma"], max_len=24, top_k=-1, top_p=g["top_p"],
                    temperature=g["temperature"], return_details=True)
    ge_off = Hh.build_product(g, DEV, graphs=True)
    off = OffloadingFlashSimpleCache(ge_off.engine.model, g["prefill"] + g["gen_len"] + 32)
    off.set_tail(g["prefill"], g["gen_len"] + 32)
    ge_off.engine.kv_cache = off
    assert ge_off.target_graphs and ge_off._target_graph(torch.zeros((1, g["gamma"] + 1), dtype=torch.long)) is None, \
        "a target graph captured over another cache must not be replayed"
    got = TriForce(tok, ge_off, prompt, gamma=g["gamma"], max_len=24, top_k=-1, top_p=g["top_p"],
                   temperature=g["temperature"], return_details=True)
    assert got["tokens"] == want["tokens"] and got["counts"] == want["counts"]
    torch.cuda.synchronize()
    S = ge_res.engine.kv_cache.seq_len
    assert off.seq_len == S
    assert torch.equal(off.cpu_k[:, :, :S], ge_res.engine.kv_cache.k[:, :, :S].cpu())      # host copy == device cache
    assert torch.equal(off.cpu_v[:, :, :S], ge_res.engine.kv_cache.v[:, :, :S].cpu())
    _, ar_off = Autoregressive(tok, ge_off, prompt, max_len=12, top_k=-1, top_p=g["top_p"], temperature=g["temperature"],
                               return_tokens=True)
    _, ar_res = Autoregressive(tok, ge_res, prompt, max_len=12, top_k=-1, top_p=g["top_p"], temperature=g["temperature"],
                               return_tokens=True)
    assert ar_off == ar_res


_RANK_STREAMS = []


def _run_concurrently(streams):
    """True when a kernel on each of ``streams`` gets to run while kernels on all the others are still busy (they sit
    on different hardware queues): every stream in turn launches a trivial kernel while the rest spin for ~20 ms."""
    import time
    flag = torch.zeros(1, device=DEV)
    for i, s in enumerate(streams):
        torch.cuda.synchronize()
        for j, o in enumerate(streams):
            if j != i:
                with torch.cuda.stream(o):
                    torch.cuda._sleep(40_000_000)                    # ~20 ms of s_sleep on one wave
        ev = torch.cuda.Event()
        with torch.cuda.stream(s):
            flag.add_(1.0)
            ev.record()
        t0 = time.time()
        while not ev.query() and time.time() - t0 < 0.008:
            pass
        ok = ev.query()
        torch.cuda.synchronize()
        if not ok:
            return False
    return True


def _rank_streams(world):
    """One stream per virtual rank, the SAME ones for every test of this process.  The kernels of the virtual ranks wait
    for each other, so they must be resident together, which needs their streams on different hardware queues; the
    runtime multiplexes streams over a handful of those, and a second pair of fresh streams has been seen to share one
    (rank 1's kernel then starts only after rank 0's READY wait has timed out).  A candidate set is therefore probed
    first (``_run_concurrently``) and replaced by fresh streams until one passes.  An artefact of several ranks on one
    device — a production rank has its device to itself."""
    if len(_RANK_STREAMS) >= world and _run_concurrently(_RANK_STREAMS[:world]):
        return _RANK_STREAMS[:world]
    for _ in range(8):
        cand = [torch.cuda.Stream(device=DEV) for _ in range(world)]
        if _run_concurrently(cand):
            _RANK_STREAMS[:] = cand
            return cand
    pytest.skip(f"no {world} streams of this process run concurrently on this device: the virtual-rank protocol tests "
                "need that (the two-process tests below cover the protocol without it)")


@pytest.mark.parametrize("alternate", [False, True], ids=["done-handshake", "alternating-halves"])
@pytest.mark.parametrize("world", [2, 4])      # one device runs 4 streams concurrently (hardware queues); 8 would serialise
def test_oneshot_allreduce_protocol_on_one_device(world, alternate):
    """tf_allreduce_oneshot with `world` virtual ranks inside this process — each rank's kernel on its own stream, real
    READY / DONE flag exchange between concurrently running kernels, fine-grained staging buffers — against the
    arithmetic it promises: fp32 accumulation in rank order, one rounding, bit-identical on every rank (at world 2 that
    is also exactly dist.all_reduce's fp16 sum).  Many back-to-back epochs on every size the decode path sends;
    no wait may ever time out."""
    from triforce_amd.utils.oneshot_ar import OneShotAllReduce, reference_sum
    hidden, max_rows = 4096, 32
    group = OneShotAllReduce.local_group(world, DEV, max_rows * hidden, alternate=alternate)
    streams = _rank_streams(world)
    gen = torch.Generator(device=DEV).manual_seed(world)
    try:
        for it, rows in enumerate([1, 7, 8, 17, 18, 32, 7, 7, 18, 1] * 3):      # three exchanges per trip: odd and even epochs
            parts = [torch.randn(rows, hidden, generator=gen, device=DEV).to(torch.float16) for _ in range(world)]
            outs = [torch.empty(rows, hidden, dtype=torch.float16, device=DEV) for _ in range(world)]
            torch.cuda.synchronize()
            for r in range(world):
                with torch.cuda.stream(streams[r]):
                    st = group[r].staging(rows, hidden)
                    st.copy_(parts[r])                               # the "producer kernel" of this rank
                    group[r].reduce(st, outs[r])
            torch.cuda.synchronize()
            want = reference_sum(parts)
            for r in range(world):
                assert torch.equal(outs[r], want), (f"epoch {it}, rank {r}: max err "
                                                    f"{(outs[r].float() - want.float()).abs().max()}, error words "
                                                    f"{[g.error() for g in group]}")
            if world == 2:
                assert torch.equal(want, parts[0] + parts[1])       # one correctly rounded fp16 addition == the ring's
            # residual form: x = x + all_reduce(partials) in the same launch, x updated in place on every rank
            xs = [torch.randn(rows, hidden, generator=gen, device=DEV).to(torch.float16) for _ in range(world)]
            want_x = [reference_sum(parts, resid=x) for x in xs]
            torch.cuda.synchronize()
            for r in range(world):
                with torch.cuda.stream(streams[r]):
                    st = group[r].staging(rows, hidden)
                    st.copy_(parts[r])
                    group[r].reduce(st, xs[r], resid=xs[r])
            torch.cuda.synchronize()
            for r in range(world):
                assert torch.equal(xs[r], want_x[r]), f"epoch {it}, rank {r}: residual form"
            # ... and with the sums of squares of the result rows per 16-column panel (the norm hand-off of the fused
            # tensor-parallel layer): same x, ss[panel][row] = sum of x[row][16 panel .. +15]^2
            ys = [x.clone() for x in want_x]
            sss = [torch.full((hidden // 16, 32), -1.0, dtype=torch.float32, device=DEV) for _ in range(world)]
            want_y = [reference_sum(parts, resid=y) for y in ys]
            torch.cuda.synchronize()
            for r in range(world):
                with torch.cuda.stream(streams[r]):
                    st = group[r].staging(rows, hidden)
                    st.copy_(parts[r])
                    group[r].reduce(st, ys[r], resid=ys[r], ss_out=sss[r])
            torch.cuda.synchronize()
            for r in range(world):
                assert torch.equal(ys[r], want_y[r])
                want_ss = want_y[r].float().square().view(rows, hidden // 16, 16).sum(-1).t()       # (panels, rows)
                Hh.close(sss[r][:, :rows], want_ss, rtol=2e-6, atol=1e-6)
                assert (sss[r][:, rows:] == -1.0).all()             # rows beyond the block are not touched
        assert [g.error() for g in group] == [0] * world
        with pytest.raises(AssertionError):
            group[0].reduce(group[0].staging(8, hidden), group[0].staging(8, hidden))     # out must not alias the staging
    finally:
        for g in group:
            g.close()


@pytest.mark.parametrize("world", [2, 4])
def test_oneshot_allreduce_alternating_halves_back_to_back_with_skewed_ranks(world):
    """What the alternating form must survive without its DONE handshake: exchanges issued back to back with no host
    synchronisation in between, fresh data every time, and the ranks out of step — each trip delays a different rank's
    stream before its producer, so the others run ahead as far as the protocol lets them (one exchange).  A rank that
    overwrote a half a slower peer was still reading would show up as a wrong sum on that peer."""
    from triforce_amd.utils.oneshot_ar import OneShotAllReduce, reference_sum
    hidden, rows, trips = 4096, 18, 120
    group = OneShotAllReduce.local_group(world, DEV, 32 * hidden, alternate=True)
    streams = _rank_streams(world)
    gen = torch.Generator(device=DEV).manual_seed(7 * world)
    parts = torch.randn(trips, world, rows, hidden, generator=gen, device=DEV).to(torch.float16)
    outs = torch.zeros(trips, world, rows, hidden, dtype=torch.float16, device=DEV)
    torch.cuda.synchronize()
    try:
        for t in range(trips):
            for r in range(world):
                with torch.cuda.stream(streams[r]):
                    if t % world == r and t % 3 != 2:
                        torch.cuda._sleep(200_000)                   # ~0.1 ms: this rank falls behind its peers
                    st = group[r].staging(rows, hidden)
                    st.copy_(parts[t, r])
                    group[r].reduce(st, outs[t, r])
        torch.cuda.synchronize()
        assert [g.error() for g in group] == [0] * world
        for t in range(trips):
            want = reference_sum([parts[t, r] for r in range(world)])
            for r in range(world):
                assert torch.equal(outs[t, r], want), f"trip {t}, rank {r}"
        # the guard: a producer that staged into the other half than the epoch selects is an error, never a silent race
        g0 = group[0]
        g0._issued += 1                                              # host count out of step with the device epoch
        out = torch.zeros(rows, hidden, dtype=torch.float16, device=DEV)
        st = g0.staging(rows, hidden)
        st.copy_(parts[0, 0])
        g0.reduce(st, out)
        torch.cuda.synchronize()
        assert g0.error() == 3 and bool(torch.isnan(out).all())
        with pytest.raises(RuntimeError, match="staging half"):
            g0.check("test")
    finally:
        for g in group:
            g.close()


def _ipc_worker(rank, world, port, q, alternate=False):
    """One rank of a 2-process group that shares ONE device: gloo carries the handle exchange and the reference
    collective, hipIpc maps the peers' fine-grained staging / control buffers, the one-shot kernels of the two processes
    talk to each other through those mappings."""
    import os
    import sys
    import traceback
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from triforce_amd.utils.oneshot_ar import OneShotAllReduce, reference_sum
        hidden = 4096
        ar = OneShotAllReduce(rank, world, "cuda:0", 32 * hidden, alternate=alternate)   # collective: exchanges the IPC handles
        gen = torch.Generator(device="cuda:0").manual_seed(100 + rank)
        ok, worst = True, 0.0
        for it, rows in enumerate([1, 7, 18, 32, 8, 7] * 4):
            part = torch.randn(rows, hidden, generator=gen, device="cuda:0").to(torch.float16)
            ring = part.clone()
            dist.all_reduce(ring, dist.ReduceOp.SUM)                       # gloo: fp16 sum of the two partials
            st = ar.staging(rows, hidden)
            st.copy_(part)
            got = ar.reduce(st, torch.empty_like(part))
            torch.cuda.synchronize()
            ok = ok and torch.equal(got, ring)                              # world 2: bit-identical to the collective
            worst = max(worst, float((got.float() - ring.float()).abs().max()))
        err = ar.error()
        dist.barrier()
        ar.close()
        q.put((rank, "ok", ok, worst, err))
        dist.destroy_process_group()
    except Exception:
        q.put((rank, "error", traceback.format_exc()))


@pytest.mark.parametrize("alternate", [False, True], ids=["done-handshake", "alternating-halves"])
def test_oneshot_allreduce_across_two_processes_through_hipipc(alternate):
    """The part the one-device protocol test cannot reach: two PROCESSES (the production arrangement, one per rank), the
    staging and control buffers exported with hipIpcGetMemHandle and mapped by the peer, flags and partials crossing the
    process boundary.  Both ranks sit on this box's single GPU (RCCL refuses two ranks on one device, so gloo carries
    the handle exchange and the reference all_reduce); results must equal the collective bit for bit and no wait may
    time out."""
    import socket
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_ipc_worker, args=(r, 2, port, q, alternate)) for r in range(2)]
    for p in procs:
        p.start()
    outs = []
    try:
        for _ in range(2):
            outs.append(q.get(timeout=300))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    for o in outs:
        assert o[1] == "ok", o[2]
    for rank, _, ok, worst, err in outs:
        assert err == 0, f"rank {rank}: a wait timed out (code {err})"
        assert ok, f"rank {rank}: result differs from the collective by up to {worst}"


def _tp2_worker(rank, world, port, q, alternate=False, extra_env=None, loop_gamma=None, golden="tp_world2", stages_only=False,
                loop_only=False):
    """One rank of the tensor-parallel engine at world size 2 with BOTH ranks on this box's single GPU: real kernels on
    each rank's head / MLP-column shard, the decode-sized all-reduces through the one-shot kernel over hipIpc mappings
    (the production path), prefill-sized ones and the token broadcast through gloo."""
    import os
    import sys
    import traceback
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK="0", TRIFORCE_AR_ALTERNATE="1" if alternate else "0")
        os.environ.update(extra_env or {})
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from triforce_amd.models.cache import StreamingLLMEvictionCache
        from triforce_amd.models.config_yarn import LlamaConfig
        from triforce_amd.models.modeling_llama_68m import LlamaForCausalLM as Draft
        from triforce_amd.models.TP_llama import DistributedLlama
        from triforce_amd.utils.decoding import TriForce_Dist
        out = {}
        if loop_only:
            return _tp2_decode_loop(rank, world, q, out, loop_gamma)
        # 1) the four forward stages against the REFERENCE engine's own world-2 logits (tests/golden/tp_world2.pt)
        g = Hh.load_golden(golden)
        tcfg = LlamaConfig.from_dict(g["tcfg"])
        gamma = g["gamma"]
        llm = DistributedLlama("unused", config=tcfg, device=DEV, local_rank=rank, world_size=world,
                               prefill=g["prefill"], gen_len=g["gen_len"], temperature=g["temperature"], top_p=g["top_p"],
                               retrieval_budget=g["budget"], retrieval_chunk_size=g["chunk"], kv_offload=True,
                               on_chip_layers=tcfg.num_hidden_layers, gamma=gamma)
        llm.init_parameters(specs.random_state_dict(g["tcfg"], g["tseed"], head_std=g["head_std"]))
        out["oneshot_stage"] = llm._ar is not None and llm._ar.alternate == alternate
        prompt = Hh.prompt_of(g).to(DEV)
        llm.reset()
        lp = llm.prefill(prompt[:, :-1])[:, -1]
        log = []
        if "topk_idx" in g:                                # the reference's own chunk selection: teacher-forced (helpers)
            from triforce_amd import ops as _ops
            undo = Hh.force_reference_selection(_ops, g["topk_idx"][rank], log)
        lb = llm.build_retrieval_cache(prompt[:, -1:])
        if "topk_idx" in g:
            undo()
            out["swapped_at_ties"] = Hh.check_selection(log)
        S = llm.kv_cache.seq_len
        vt = torch.tensor([[11, 12, 13] + [100] * (gamma - 2)], device=DEV)
        ls = llm.retrieval_inference(vt, torch.arange(S, S + gamma + 1, device=DEV).unsqueeze(0))
        lv = llm.inference(vt)
        torch.cuda.synchronize()
        out.update(S=S, stages=[t.float().cpu().numpy() for t in (lp, lb, ls, lv)],      # by value (see tests/test_tp_cpu.py)
                   ar_error_stage=llm._ar.error() if llm._ar is not None else -1,
                   xchg=llm._xchg is not None, xchg_form=getattr(llm, "xchg_form", None), note=getattr(llm, "allreduce_note", ""))
        if stages_only:
            dist.barrier()
            q.put((rank, "ok", out))
            dist.destroy_process_group()
            return
        del llm
        _tp2_decode_loop(rank, world, q, out, loop_gamma)
    except Exception:
        q.put((rank, "error", traceback.format_exc()))


def _tp2_decode_loop(rank, world, q, out, loop_gamma=None):
    """2) the whole decode loop (draft + retrieval verify + target verify, hipGraphs) on the small_gamma6 fixture;
    $TF_TEST_TP_SAMPLING = "T,top_p" replaces the fixture's greedy target by a stochastic one."""
    import os
    import torch.distributed as dist
    from triforce_amd.models.cache import StreamingLLMEvictionCache
    from triforce_amd.models.config_yarn import LlamaConfig
    from triforce_amd.models.modeling_llama_68m import LlamaForCausalLM as Draft
    from triforce_amd.models.TP_llama import DistributedLlama
    from triforce_amd.utils.decoding import TriForce_Dist
    g = Hh.load_golden("small_gamma6")
    if os.environ.get("TF_TEST_TP_SAMPLING"):
        g["temperature"], g["top_p"] = (float(x) for x in os.environ["TF_TEST_TP_SAMPLING"].split(","))
    gamma = loop_gamma or g["gamma"]
    max_len = int(os.environ.get("TF_TEST_TP_MAXLEN", "24"))
    if max_len > 24:
        # a long run: the retrieval cache's tail holds every generated token (reference cache.py:180-182), so the budget — and
        # with it the prompt — must be longer than the run: 4 096-token prompt, budget = run length + slack (multiple of the chunk)
        g["prefill"], g["gen_len"] = 4096, max_len + 64
        g["budget"] = min(4096 - 512, ((max_len + 128 + 7) // 8) * 8)
        g["tcfg"] = dict(g["tcfg"], max_position_embeddings=8192)        # RoPE tables must cover prompt + generated positions
    draft = Draft.from_state_dict(LlamaConfig.from_dict(g["dcfg"]),
                                  specs.random_state_dict(g["dcfg"], g["dseed"], head_std=g["head_std"]), DEV)
    dcache = StreamingLLMEvictionCache(draft, start_size=16, recent_size=256 - 16 - gamma, gamma=gamma)
    tcfg = LlamaConfig.from_dict(g["tcfg"])
    llm = DistributedLlama("unused", config=tcfg, device=DEV, local_rank=rank, world_size=world,
                           prefill=g["prefill"], gen_len=g["gen_len"], temperature=g["temperature"], top_p=g["top_p"],
                           retrieval_budget=g["budget"], kv_offload=True, on_chip_layers=tcfg.num_hidden_layers,
                           draft=draft, draft_cache=dcache, gamma=gamma)
    llm.init_parameters(specs.random_state_dict(g["tcfg"], g["tseed"], head_std=g["head_std"]))
    llm.initialize_graphs()
    tok = Hh.FakeTokenizer()
    if max_len > 24:
        tok.eos_token_id = 1 << 30                         # a long run: no id ends it (a random model samples id 2 within ~200 tokens)
    try:
        res = TriForce_Dist(tok, llm, Hh.prompt_of(g).to(DEV), gamma=gamma, max_len=max_len, top_k=-1,
                            top_p=g["top_p"], temperature=g["temperature"], return_details=True)
    except RuntimeError as ex:                             # a loop that failed LOUDLY (stream digest / exchange time-out)
        if not os.environ.get("TF_TEST_TP_EXPECT_FAILURE"):
            raise
        out.update(failed=f"{type(ex).__name__}: {ex}"[:400])
        q.put((rank, "ok", out))
        q.close()
        q.join_thread()                                    # (flush before the hard exit)
        os._exit(0)                                        # (the group is out of step: no barrier, no orderly shutdown)
    torch.cuda.synchronize()
    out.update(tokens=res["tokens"], counts=res["counts"], seq_len=llm.kv_cache.seq_len, graph_form=llm.graph_form,
               decisions=res["decisions"], replica_checks=res["replica_checks"], inner_graphs=res.get("inner_graphs"),
               oneshot_decode=llm._ar is not None, ar_error_decode=llm._ar.error() if llm._ar is not None else -1)
    dist.barrier()
    q.put((rank, "ok", out))
    dist.destroy_process_group()


@pytest.mark.parametrize("alternate", [False, True], ids=["done-handshake", "alternating-halves"])
def test_tp_world2_on_one_device_real_kernels_and_oneshot_allreduce(alternate):
    """World size 2 on hardware, as far as a 1-GPU box allows: two processes, each with its own shard of the heads and
    of the MLP columns, running the real kernels on the same device; every decode-sized all-reduce is the one-shot
    kernel across the process boundary.  (1) the four forward stages equal the logits the UNMODIFIED reference engine
    produced as 2 gloo processes (same tolerance as the CPU test of the host logic), bit-identical on both ranks;
    (2) the graphed decode loop emits the same stream on both ranks and stays on the oracle's greedy path."""
    import socket
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_tp2_worker, args=(r, 2, port, q, alternate)) for r in range(2)]
    for p in procs:
        p.start()
    outs = {}
    try:
        for _ in range(2):
            o = q.get(timeout=600)
            outs[o[0]] = o
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    for o in outs.values():
        assert o[1] == "ok", o[2]
    a, b = outs[0][2], outs[1][2]
    g = Hh.load_golden("tp_world2")
    for r, o in ((0, a), (1, b)):
        assert o["oneshot_stage"] and o["oneshot_decode"], f"rank {r}: the engine fell back to the ring collective"
        assert o["ar_error_stage"] == 0 and o["ar_error_decode"] == 0, f"rank {r}: an all-reduce wait timed out"
        assert o["S"] == g["S"]
        o["stages"] = [torch.from_numpy(x) for x in o["stages"]]
        for name, ours in zip(("prefill_logits", "build_logits", "spec_logits", "verify_logits"), o["stages"]):
            gap = (ours.reshape(g[name].shape) - g[name].float()).abs().max().item()
            assert Hh.bound("rank logits vs the reference's world-2 golden logits", gap, 4e-3), f"rank {r} {name}: {gap:.2e} from the reference's world-2 logits"
    for x, y in zip(a["stages"], b["stages"]):
        assert torch.equal(x, y)                                  # every rank holds the same bits after each all-reduce
    assert a["tokens"] == b["tokens"] and a["counts"] == b["counts"] and a["seq_len"] == b["seq_len"]
    print(f"[tp2 on one device] graph form {a['graph_form']}, {len(a['tokens'])} tokens, accept counts {a['counts']}")
    g6 = Hh.load_golden("small_gamma6")
    gaps = Hh.teacher_forced_gaps(g6, a["tokens"])
    assert max(gaps) < 8e-3, f"TP stream leaves the oracle's greedy path: gap {max(gaps):.4f}"
    assert Hh.common_prefix(a["tokens"], g6["ar_tokens"]) >= 12


def _run_tp2(**kw):
    import socket
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_tp2_worker, args=(r, 2, port, q), kwargs=kw) for r in range(2)]
    for p in procs:
        p.start()
    outs = {}
    try:
        for _ in range(2):
            o = q.get(timeout=300)
            outs[o[0]] = o
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    for o in outs.values():
        assert o[1] == "ok", o[2]
    return outs[0][2], outs[1][2]


def test_tp_world2_replicated_decisions_emit_the_broadcast_stream_on_the_device():
    """TRIFORCE_TP_REPLICATED_DECISIONS=1 with the real kernels (two processes on this GPU, the exchange across the process
    boundary, whole-forward hipGraphs): no record broadcast, the records through each rank's pinned mailbox — both ranks emit
    the broadcast form's stream, accept counts and cache length under a stochastic target (T = 0.8, top-p 0.9), the stream
    digest agreed across the ranks at every check, no exchange error."""
    env = {"TF_TEST_TP_SAMPLING": "0.8,0.9"}
    base, _ = _run_tp2(extra_env=dict(env, TRIFORCE_TP_REPLICATED_DECISIONS="0"), loop_only=True)
    a, b = _run_tp2(extra_env=dict(env, TRIFORCE_TP_REPLICATED_DECISIONS="1", TRIFORCE_TP_REPLICA_CHECK_EVERY="2"), loop_only=True)
    assert base["decisions"] == "broadcast" and a["decisions"] == b["decisions"] == "replicated"
    for r, o in ((0, a), (1, b)):
        assert o["tokens"] == base["tokens"] and o["counts"] == base["counts"] and o["seq_len"] == base["seq_len"], \
            f"rank {r} left the broadcast form's stream"
        assert o["replica_checks"] >= len(o["counts"]) // 2 and o["ar_error_decode"] == 0 and o["oneshot_decode"]
        # round 6: with whole-forward graphs the replicated loop runs the single-GPU loop's structure — one hipGraph per inner
        # iteration (draft step, draw, retrieval verify WITH its exchanges, accept test), records through the pinned mailbox
        assert o["inner_graphs"] == (o["graph_form"] == "whole"), (o["inner_graphs"], o["graph_form"])
    assert not base["inner_graphs"]                                  # a broadcast cannot sit inside an iteration's graph
    assert len(set(base["tokens"])) > 4
    print(f"[tp2 replicated decisions] {len(a['tokens'])} tokens, accept counts {a['counts']}, {a['replica_checks']} digest checks")


@pytest.mark.parametrize("xchg", ["1", "0"], ids=["exchange-in-gemm", "staged-exchange"])
def test_tp_world2_replicated_decisions_soak_2000_tokens(xchg):
    """The evidence behind the "auto" default (utils/decoding.tp_sync_record): two processes on this GPU, T = 1.0 / top-p 0.95 (every
    decision draws numbers), 2 000+ tokens, in both exchange forms (inside the o_proj / down_proj GEMMs; GEMM -> staging ->
    exchange kernel).  With NO setting the start-up litmus must find the ranks' forwards bit-identical and select replicated
    decisions; both ranks then emit, token for token, the stream of the forced broadcast form, with the same accept counts and
    cache length, every digest check agreeing and no exchange error."""
    env = {"TF_TEST_TP_SAMPLING": "1.0,0.95", "TF_TEST_TP_MAXLEN": "2000", "TRIFORCE_TP_GEMM_XCHG": xchg,
           "TRIFORCE_TP_REPLICA_CHECK_EVERY": "16"}
    base, base1 = _run_tp2(extra_env=dict(env, TRIFORCE_TP_REPLICATED_DECISIONS="0"), loop_only=True)
    a, b = _run_tp2(extra_env=env, loop_only=True)                               # default: auto
    assert base["decisions"] == "broadcast" and a["decisions"] == b["decisions"] == "replicated", (base["decisions"], a["decisions"])
    assert len(base["tokens"]) >= 2000 and base["tokens"] == base1["tokens"]
    for r, o in ((0, a), (1, b)):
        assert o["tokens"] == base["tokens"] and o["counts"] == base["counts"] and o["seq_len"] == base["seq_len"], \
            f"rank {r} left the broadcast form's stream after {Hh.common_prefix(o['tokens'], base['tokens'])} tokens"
        assert o["replica_checks"] >= len(o["counts"]) // 16 and o["ar_error_decode"] == 0 and o["oneshot_decode"]
        assert o["inner_graphs"] == (o["graph_form"] == "whole")
    assert len(set(base["tokens"])) > 100
    Hh.note(f"tp2 replicated-decision soak ({'exchange in the GEMMs' if xchg == '1' else 'staged exchange'}): {len(a['tokens'])} tokens / "
            f"{len(a['counts'])} outer steps identical to the broadcast form, {a['replica_checks']} digest checks, graph form {a['graph_form']}")


def test_tp_world2_replicated_decisions_fail_loudly_when_a_rank_leaves_the_stream():
    """Rank 1's uniform stream is knocked one number out of step at outer step 6 (TF_TEST_TP_KNOCK_RANK_AT): its decisions now
    differ from rank 0's.  Replicated decisions have no broadcast to paper over that — the job must STOP, on every rank, within
    the digest interval: either the stream digests disagree (ReplicaCheck) or the ranks' forwards fall out of step and an
    exchange times out.  What must not happen is two ranks finishing with two streams."""
    env = {"TF_TEST_TP_SAMPLING": "1.0,0.95", "TF_TEST_TP_MAXLEN": "400", "TRIFORCE_TP_REPLICATED_DECISIONS": "1",
           "TRIFORCE_TP_REPLICA_CHECK_EVERY": "8", "TF_TEST_TP_KNOCK_RANK_AT": "1,6", "TF_TEST_TP_EXPECT_FAILURE": "1",
           "TRIFORCE_XCHG_TIMEOUT_MS": "1500"}
    a, b = _run_tp2(extra_env=env, loop_only=True)
    for r, o in ((0, a), (1, b)):
        assert "failed" in o, f"rank {r} finished {len(o.get('tokens', []))} tokens as if nothing had happened"
    why = a["failed"] + " | " + b["failed"]                # (the second rank to notice may only see its peer gone)
    assert "left the common token stream" in why or "timed out" in why or "time-out" in why or "exchange" in why, why
    Hh.note(f"tp2 knocked rank: rank 0 stopped with [{a['failed'][:90]}], rank 1 with [{b['failed'][:90]}]")


def test_tp_world8_on_one_device_matches_the_reference_world8_logits():
    """BASELINE configs[4]'s world size on hardware, as far as a one-GPU box allows (round 5; verdict round 4, item 4b):
    EIGHT processes, one attention head and 128 MLP columns each, real kernels, the decode-sized all-reduces through the fused
    GEMM + exchange across seven process boundaries (hipIpc mappings; its start-up litmus runs on this very group and
    its verdict is part of the result), prefill-sized ones through gloo.  The four forward stages against the logits the
    UNMODIFIED reference engine produced as 8 gloo processes (tests/golden/tp_world8.pt, oracle/gen_golden.py tp8), with the
    reference's per-rank chunk selection teacher-forced and the product's own selection judged tie-tolerantly; every rank
    must hold the same bits.  (The reference's exchange: dist.all_reduce, models/tensor_op.py:179,326,359.)"""
    import socket
    import torch.multiprocessing as mp
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {"TRIFORCE_XCHG_LITMUS_ITERS": "400"}             # (8 ranks time-share ONE device here: every exchange needs all of them scheduled)
    procs = [ctx.Process(target=_tp2_worker, args=(r, world, port, q, False, env, None, "tp_world8", True)) for r in range(world)]
    for p in procs:
        p.start()
    outs = {}
    try:
        for _ in range(world):
            o = q.get(timeout=900)
            outs[o[0]] = o
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    for o in outs.values():
        assert o[1] == "ok", o[2]
    g = Hh.load_golden("tp_world8")
    res = [outs[r][2] for r in range(world)]
    for r, o in enumerate(res):
        assert o["oneshot_stage"], f"rank {r}: the engine fell back to the ring collective ({o['note']})"
        assert o["xchg"] and "litmus" in (o["xchg_form"] or ""), f"rank {r}: fused exchange not selected: {o['xchg_form']} / {o['note']}"
        assert o["ar_error_stage"] == 0 and o["S"] == g["S"]
        o["stages"] = [torch.from_numpy(x) for x in o["stages"]]
        for name, ours in zip(("prefill_logits", "build_logits", "spec_logits", "verify_logits"), o["stages"]):
            gap = (ours.reshape(g[name].shape) - g[name].float()).abs().max().item()
            assert Hh.bound("rank logits vs the reference's world-8 golden logits", gap, 8e-3), f"rank {r} {name}: {gap:.2e} from the reference's world-8 logits"
    for o in res[1:]:
        for x, y in zip(res[0]["stages"], o["stages"]):
            assert torch.equal(x, y)                              # every rank holds the same bits after each all-reduce
    Hh.note(f"tp world 8 on one device: exchange {res[0]['xchg_form']}; chunks swapped at ties "
            f"{sum(o.get('swapped_at_ties', 0) for o in res)}; max gaps vs the reference's world-8 logits "
            + ", ".join(f"{(x.reshape(g[n].shape) - g[n].float()).abs().max().item():.2e}" for n, x in
                        zip(("prefill_logits", "build_logits", "spec_logits", "verify_logits"), res[0]["stages"])))


@pytest.mark.parametrize("fuse", ["1", "0"], ids=["fused-layer", "unfused-layer"])
def test_tp_world2_gamma16_segment_graphs_with_alternating_halves(fuse):
    """The advisor's round-3 case: alternating staging halves + the collective-free SEGMENT graphs + a gamma = 16 verify
    (17 / 18 rows), in the fused layer (k-octet-major activations) and in the un-fused one, whose two partials per layer
    used to be taken from staging() once — both in the same half.  Every stage must be captured against the half its
    exchange reads at replay; any disagreement is the sticky error 3 with NaN output.  Both ranks must stay on the
    oracle's greedy path with error word 0."""
    import socket
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {"TRIFORCE_TP_GRAPHS": "segments", "TRIFORCE_TP_FUSE": fuse, "TRIFORCE_TP_GEMM_XCHG": "0"}
    procs = [ctx.Process(target=_tp2_worker, args=(r, 2, port, q, True, env, 16)) for r in range(2)]
    for p in procs:
        p.start()
    outs = {}
    try:
        for _ in range(2):
            o = q.get(timeout=600)
            outs[o[0]] = o
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    for o in outs.values():
        assert o[1] == "ok", o[2]
    a, b = outs[0][2], outs[1][2]
    for r, o in ((0, a), (1, b)):
        assert o["oneshot_decode"] and o["graph_form"] == "segments", (r, o["graph_form"])
        assert o["ar_error_stage"] == 0 and o["ar_error_decode"] == 0, f"rank {r}: exchange error {o['ar_error_decode']}"
    assert a["tokens"] == b["tokens"] and a["counts"] == b["counts"]
    g6 = Hh.load_golden("small_gamma6")
    gaps = Hh.teacher_forced_gaps(g6, a["tokens"])
    assert max(gaps) < 8e-3, f"stream leaves the oracle's greedy path: gap {max(gaps):.4f}"


@pytest.mark.parametrize("world,hidden", [(2, 4096), (4, 1024)])
def test_gemm_exchange_protocol_on_one_device(world, hidden):
    """tf_skinny_gemm_xchg — o_proj / down_proj with the all-reduce inside the GEMM's epilogue — with `world` virtual ranks
    in this process, one stream each: every workgroup publishes its panel, flags the same panel of the peers' concurrently
    running kernels, reads theirs.  Must equal, bit for bit, GEMM -> fp16 partial -> fp32 sum in rank order -> one
    rounding -> fp16 residual add (utils.oneshot_ar.reference_sum of the ranks' ops.linear outputs), in both activation
    layouts, back to back over odd and even epochs with no host synchronisation in between, one rank delayed each trip.
    (4 ranks run at hidden 1024 = 64 workgroups each: the grids of all virtual ranks must be resident together.)"""
    from triforce_amd import ops
    from triforce_amd.utils.oneshot_ar import GemmExchange, reference_sum
    group = GemmExchange.local_group(world, DEV, 32 * hidden)
    streams = _rank_streams(world)
    gen = torch.Generator(device=DEV).manual_seed(7 + world)
    K = 512
    ws = [ops.PackedLinear((torch.randn(hidden, K, generator=gen, device=DEV) * 0.05).to(torch.float16)) for _ in range(world)]
    try:
        for it, rows in enumerate([1, 7, 17, 18, 32, 8, 7] * 2):
            packed = it % 2 == 1
            acts = [torch.randn(rows, K, generator=gen, device=DEV).to(torch.float16) for _ in range(world)]
            resid = torch.randn(rows, hidden, generator=gen, device=DEV).to(torch.float16)
            want = reference_sum([ops.linear(a, w) for a, w in zip(acts, ws)], resid=resid)
            xs = [ops.Act.from_rows(resid) if packed else resid.clone() for _ in range(world)]
            sss = [ops.ss_buffer(hidden, DEV) for _ in range(world)]
            ins = [ops.Act.from_rows(a) if packed else a for a in acts]
            torch.cuda.synchronize()
            for r in range(world):
                with torch.cuda.stream(streams[r]):
                    if it % world == r:
                        torch.cuda._sleep(100_000)                        # this rank arrives late
                    group[r].linear_reduce(ins[r], ws[r], xs[r], sss[r])
                    group[r].linear_reduce(ins[r], ws[r], xs[r], sss[r])  # back to back: the other staging half
            torch.cuda.synchronize()
            want2 = reference_sum([ops.linear(a, w) for a, w in zip(acts, ws)], resid=want)
            for r in range(world):
                got = xs[r].rows() if packed else xs[r]
                assert torch.equal(got, want2), (f"trip {it}, rank {r}, {rows} rows, packed={packed}: max err "
                                                 f"{(got.float() - want2.float()).abs().max()}, errors {[g.error() for g in group]}")
                Hh.close(sss[r][:, :rows], want2.float().square().view(rows, hidden // 16, 16).sum(-1).t(),
                                           rtol=2e-6, atol=1e-6)
        assert [g.error() for g in group] == [0] * world and [g.error_device() for g in group] == [0] * world
        # error path: a poisoned control block NaN-fills the output and check() raises
        group[0].inject_error(1)
        x = torch.zeros(7, hidden, dtype=torch.float16, device=DEV)
        with torch.cuda.stream(streams[0]):
            group[0].linear_reduce(torch.randn(7, K, generator=gen, device=DEV).to(torch.float16), ws[0], x,
                                   ops.ss_buffer(hidden, DEV))
        torch.cuda.synchronize()
        assert bool(torch.isnan(x).all())
        with pytest.raises(RuntimeError, match="GEMM \\+ exchange"):
            group[0].check("test")
    finally:
        for g in group:
            g.close()


def _xchg_ipc_worker(rank, world, port, q):
    """One rank of a 2-process group on ONE device: GemmExchange buffers exported / mapped through hipIpc, the GEMM kernels of
    the two processes exchanging their panels through those mappings; against gloo's all-reduce of the fp16 partials."""
    import os
    import sys
    import traceback
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from triforce_amd import ops
        from triforce_amd.utils.oneshot_ar import GemmExchange
        hidden, K = 4096, 512
        xc = GemmExchange(rank, world, "cuda:0", 32 * hidden)
        gen = torch.Generator(device="cuda:0").manual_seed(200 + rank)
        w = ops.PackedLinear((torch.randn(hidden, K, generator=gen, device="cuda:0") * 0.05).to(torch.float16))
        ok, worst = True, 0.0
        for it, rows in enumerate([1, 7, 18, 32, 8, 7] * 3):
            a = torch.randn(rows, K, generator=gen, device="cuda:0").to(torch.float16)
            ring = ops.linear(a, w)
            dist.all_reduce(ring, dist.ReduceOp.SUM)                       # gloo: fp16 sum of the two partials
            x = torch.zeros(rows, hidden, dtype=torch.float16, device="cuda:0")
            xc.linear_reduce(a, w, x, ops.ss_buffer(hidden, "cuda:0"))
            torch.cuda.synchronize()
            ok = ok and torch.equal(x, ring)
            worst = max(worst, float((x.float() - ring.float()).abs().max()))
        err = xc.error()
        dist.barrier()
        xc.close()
        q.put((rank, "ok", ok, worst, err))
        dist.destroy_process_group()
    except Exception:
        q.put((rank, "error", traceback.format_exc()))


def test_gemm_exchange_across_two_processes_through_hipipc():
    import socket
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_xchg_ipc_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = []
    try:
        for _ in range(2):
            outs.append(q.get(timeout=300))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    for o in outs:
        assert o[1] == "ok", o[2]
    for rank, _, ok, worst, err in outs:
        assert err == 0, f"rank {rank}: a panel wait timed out (code {err})"
        assert ok, f"rank {rank}: result differs from the collective by up to {worst}"


def _run_litmus(nproc, extra):
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(root, "tools", "xgmi_litmus.py"), *extra]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, lines, p.stderr[-2000:]


@pytest.mark.parametrize("alternate", [False, True], ids=["done-handshake", "alternating-halves"])
def test_exchange_litmus_dry_run_two_processes_one_device(alternate):
    """tools/xgmi_litmus.py as two processes sharing this box's one GPU (gloo + hipIpc mappings): plain-store producer
    kernel -> kernel boundary -> READY flag -> peer's fine-grained loads, eager and inside a hipGraph, one rank delayed
    now and then.  On one device the mappings resolve to local HBM, so this is a dry run of the tool and of the
    protocol across processes; `bash tools/gpu_validate.sh tp N` runs it across N devices before the bench legs."""
    rc, lines, err = _run_litmus(2, ["--share-device", "--eager-iters", "3000", "--iters", "40000", "--budget-s", "40"]
                                 + (["--alternate"] if alternate else []))
    assert rc == 0 and len(lines) == 1, err
    j = lines[0]
    assert j["litmus_ok"] and j["failures"] == 0 and j["world"] == 2
    assert j["eager"]["iterations"] == 3000 and j["graph"]["iterations"] >= 2000
    assert all(e["error_word"] == 0 and e["mismatched_elements"] == 0 for e in j["per_rank"])


def test_rccl_world2_and_exchange_litmus_across_two_devices():
    """The first N > 1 execution on a multi-GPU box: RCCL initialised at world size 2 (one process per device), the
    exchange litmus across the two devices (xGMI remote loads and flag stores).  Skipped on a one-GPU box."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible devices")
    rc, lines, err = _run_litmus(2, ["--eager-iters", "5000", "--iters", "200000", "--budget-s", "40"])
    assert rc == 0 and len(lines) == 1, err
    assert lines[0]["litmus_ok"] and lines[0]["failures"] == 0 and not lines[0]["share_device"]


def test_fused_exchange_litmus_both_forms_two_processes_one_device():
    """The litmus of the FUSED exchange itself (round 5; the staged exchange kernel has had one since round 4): two
    processes on this box's GPU, `tools/xgmi_litmus.py --xchg` = utils.oneshot_ar.GemmExchange.litmus — fresh integer
    pattern -> tf_skinny_gemm_xchg over an identity weight -> every element of the sum checked against the SAME iteration's
    patterns, inside replayed hipGraphs, one rank delayed now and then — first the fence-free form, then (control blocks
    reset collectively) the fenced one.  The engine runs exactly this on the real group at start-up and keeps the
    fence-free form only if it is clean (models/TP_llama.py _xchg_litmus)."""
    rc, lines, err = _run_litmus(2, ["--share-device", "--xchg", "--iters", "20000", "--per-graph", "100"])
    assert rc == 0 and len(lines) == 1, err
    j = lines[0]
    assert j["litmus_ok"] and j["failures"] == 0 and j["kernel"] == "tf_skinny_gemm_xchg"
    assert [f["fenced"] for f in j["forms"]] == [False, True]
    for f in j["forms"]:
        assert f["iterations"] >= 20000 and f["mismatched_elements"] == 0 and f["error_word"] == 0
        assert all(e["error_word"] == 0 and e["mismatched_elements"] == 0 for e in f["per_rank"])


def test_fused_exchange_litmus_across_two_devices():
    """... across two devices (xGMI remote loads, remote flag stores).  Skipped on a one-GPU box."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible devices")
    rc, lines, err = _run_litmus(2, ["--xchg", "--iters", "200000"])
    assert rc == 0 and len(lines) == 1, err
    assert lines[0]["litmus_ok"] and not lines[0]["share_device"]


def test_fused_exchange_timeout_is_wall_clock_and_reset_recovers():
    """Advisor, round 4: a peer that never flags must be reported in SECONDS (wall clock, tf_xchg_tune key 1), not after
    2^27 polls, and the group must be able to start over (tf_xchg_reset: per-panel counts, flags, error word).  Two virtual
    ranks on this device; rank 1 simply does not launch: rank 0's exchange times out inside the limit, NaN-fills its panels
    and sets the error word (+ host mirror); after a reset of both control blocks the pair exchanges correctly again."""
    import time
    from triforce_amd import ops
    from triforce_amd.utils.oneshot_ar import GemmExchange
    hidden, K, rows = 1024, 256, 7
    dev = torch.device(DEV)
    grp = GemmExchange.local_group(2, dev, 32 * hidden)
    old = GemmExchange.set_timeout_ms(300)
    try:
        gen = torch.Generator(device=dev).manual_seed(5)
        w = ops.PackedLinear((torch.randn(hidden, K, generator=gen, device=dev) * 0.05).to(torch.float16))
        acts = [torch.randn(rows, K, generator=gen, device=dev).to(torch.float16) for _ in range(2)]
        x0 = torch.zeros(rows, hidden, dtype=torch.float16, device=dev)
        t0 = time.time()
        grp[0].linear_reduce(acts[0], w, x0, ops.ss_buffer(hidden, dev))          # the peer never arrives
        torch.cuda.synchronize()
        waited = time.time() - t0
        assert 0.2 < waited < 5.0, f"time-out after {waited:.2f} s (limit 0.3 s)"
        assert grp[0].error() == 1 and grp[0].error_device() == 1 and bool(torch.isnan(x0.float()).any())
        with pytest.raises(RuntimeError):
            grp[0].check("test")
        for g in grp:                                                              # collective by contract: nothing in flight
            g.reset()
        assert grp[0].error() == 0 and grp[0].error_device() == 0
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        outs = [torch.zeros(rows, hidden, dtype=torch.float16, device=dev) for _ in range(2)]
        for it in range(3):                                                        # odd / even epochs: both staging halves
            for o in outs:
                o.zero_()
            torch.cuda.synchronize()
            for r in range(2):
                with torch.cuda.stream(streams[r]):
                    grp[r].linear_reduce(acts[r], w, outs[r], ops.ss_buffer(hidden, dev))
            torch.cuda.synchronize()
            want = (ops.linear(acts[0], w).float() + ops.linear(acts[1], w).float()).half()
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], want), it
        assert grp[0].error() == 0 and grp[1].error() == 0
    finally:
        GemmExchange.set_timeout_ms(old)
        for g in grp:
            g.close()


def test_oneshot_allreduce_error_path_poisons_output_and_raises():
    """A timed-out exchange must never pass for a result: with the sticky error word set (fault injection — what a
    READY / DONE timeout leaves behind) every later reduce returns at once with `out` NaN-filled, `check()` raises, and
    after the word is cleared the same group reduces correctly again."""
    from triforce_amd.utils.oneshot_ar import OneShotAllReduce
    hidden = 512
    ar = OneShotAllReduce.local_group(1, DEV, 32 * hidden)[0]
    part = torch.randn(7, hidden, device=DEV).to(torch.float16)
    st = ar.staging(7, hidden)
    st.copy_(part)
    out = ar.reduce(st, torch.zeros_like(part))
    torch.cuda.synchronize()
    assert torch.equal(out, part) and ar.error() == 0
    ar.check()
    ar.inject_error(1)
    out = ar.reduce(st, torch.zeros_like(part))
    resid = torch.ones_like(part)
    ss = torch.zeros(hidden // 16, 32, dtype=torch.float32, device=DEV)
    out2 = ar.reduce(st, resid, resid=resid, ss_out=ss)
    torch.cuda.synchronize()
    assert bool(torch.isnan(out).all()) and bool(torch.isnan(out2).all()), "error path left plausible values in out"
    assert ar.error() == 1
    with pytest.raises(RuntimeError, match="timed out"):
        ar.check("unit test")
    ar.inject_error(0)
    st.copy_(part)
    out = ar.reduce(st, torch.zeros_like(part))
    torch.cuda.synchronize()
    assert torch.equal(out, part) and ar.error() == 0
    ar.close()


def _bench_tp2(extra_env, extra_args=()):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # the tiny target has 2 layers: time EVERY eager attention launch (the default samples every 8th of 32 per verify)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", TRIFORCE_ATTN_TIMER_EVERY="1", **extra_env)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-device", "--target", "tiny", "--prefill",
           "2048", "--budget", "256", "--gamma", "4", "--steps", "4", "--warmup", "1", "--weights", "random",
           "--no-cpu-baseline", "--roofline-every", "1", *extra_args]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, lines, p.stderr[-2000:]


def test_bench_tp_line_reports_allreduce_state_and_fails_loudly_on_a_timeout():
    """bench.py --gpus 2 (two ranks sharing this box's one GPU; gloo carries the large collectives, the decode-sized ones
    are the one-shot kernel across the process boundary).  Healthy run: rc 0, the line says which exchange and which
    graph form ran, allreduce_error 0, and carries a roofline from the eagerly sampled target verifies.  With a timeout
    injected on every rank after warm-up: rc != 0 and a JSON line with `failed` and the error word — never tokens/s."""
    rc, lines, err = _bench_tp2({}, ("--allreduce", "oneshot", "--require-graph-form", "whole"))
    assert rc == 0 and len(lines) == 1, err
    j = lines[0]
    assert j["allreduce_error"] == 0 and j["decode_allreduce"].startswith("one-shot") and j["graph_form"] == "whole"
    assert j["config"]["world_size_observed"] == 2 and j["value"] > 0
    assert j["roofline"] is not None and j["roofline"]["launches"] >= 1, "no eager target verify was sampled"
    assert j["config"]["workload"].startswith("custom")
    # round 5: the N > 1 line carries what the scaling prediction is written in — per-rank stage latencies measured by all
    # ranks in lock-step, the exchange form selected (with its litmus verdict), exchanges per step, a measured per-exchange time
    mr = j["multi_rank"]
    assert [r["rank"] for r in mr["stage_latency_us_per_rank"]] == [0, 1]
    for r in mr["stage_latency_us_per_rank"]:
        assert r["draft_step_us"] > 0 and r["retrieval_verify_us"] > 0 and r["target_verify_us"] > 0
    assert mr["stage_latency_us_slowest_rank"]["target_verify_us"] == max(r["target_verify_us"] for r in mr["stage_latency_us_per_rank"])
    ex = mr["exchange"]
    assert ex["per_forward"] == 4 and ex["per_step"] > 4 and "tf_skinny_gemm_xchg" in ex["form"] and "litmus" in ex["form"]
    for r in ex["per_rank"]:
        assert "failed" not in r, r
        assert r["gemm_with_exchange_us"] > 0 and r["gemm_alone_us"] > 0 and "per_exchange_us" in r, r
    assert set(mr["measured_step_terms_us"]) >= {"target_verify", "retrieval_verify", "draft", "host_and_broadcasts"}
    assert j["stage_latency_us"] == {k: v for k, v in mr["stage_latency_us_per_rank"][0].items() if k != "rank"}
    rc, lines, err = _bench_tp2({"TRIFORCE_BENCH_INJECT_AR_ERROR": "1"}, ("--allreduce", "oneshot"))
    assert rc != 0, "a timed-out all-reduce must fail the bench"
    bad = [ln for ln in lines if ln.get("failed")]
    assert bad and all(ln["value"] is None and ln["allreduce_error"] == 1 for ln in bad), (lines, err)
    assert not any(ln.get("value") for ln in lines)
