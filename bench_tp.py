"""bench.py --gpus N (N>1): tensor-parallel TriForce decode, one process per GPU over RCCL/xGMI.

Same model, prompt length and decode loop as the 1-GPU workload; attention heads, MLP columns, the KV cache
and the retrieval cache are sharded N ways, with the two fp16 all-reduces per layer as the only
exchange step.  Total work is fixed, so scaling is "strong".  All layers stay resident in HBM (on_chip = L):
with 288 GB per MI355X the offloading tier is never needed for these shapes (it is exercised by the tests and
by test/offloading_TP.py --on_chip)."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist


def offload_report(llm, args, tcfg, world, device):
    """The offloading tier in numbers (this rank): bytes a target verify pulls over PCIe, its latency, the H2D rate that
    implies, the rate of the same copies with nothing else running, and how much of the shorter of (copy, compute) is
    hidden under the longer one."""
    from triforce_amd import ops
    L, n_on = tcfg.num_hidden_layers, llm.on_chip_layers
    S = llm.kv_cache.seq_len
    q = args.gamma + 2
    ids = torch.full((1, q), 100, dtype=torch.long, device=device)

    def timed(fn, n=2):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n

    def verify():
        llm.inference(input_ids=ids)
        llm.kv_cache.seq_len = S                               # probe: roll back (the written-back rows are scratch)
    t_verify = timed(verify)
    Hl, D = tcfg.num_attention_heads // world, tcfg.head_dim
    per_layer = 2 * S * Hl * D * 2
    h2d = per_layer * (L - n_on)

    def pure_copy():                                           # one offloaded layer's K and V, copy stream only
        llm.kv_buffer[0].copy_kv(llm.kv_cache, n_on, llm.load_stream)
        torch.cuda.current_stream().wait_stream(llm.load_stream)
    t_copy_layer = timed(pure_copy, n=3)
    pure_rate = per_layer / t_copy_layer / 1e6                 # GB/s
    t_copy = t_copy_layer * (L - n_on)
    # compute alone: the same forward over HBM-resident layers only, scaled to all layers
    llm_on = max(n_on, 1)
    t_compute = None
    try:
        x = llm.embed_tokens[ids.reshape(-1)]
        pos = (S + torch.arange(q, device=device)).contiguous()

        def resident():
            d = None
            for idx in range(llm_on):
                kl, vl = llm.kv_cache.layer_kv(idx)
                d = llm._layer(idx, x.clone(), d, pos, kl, vl, S, S + q, q)
        t_compute = timed(resident) * L / llm_on
    except Exception:
        pass
    out = {"on_chip_layers": n_on, "offloaded_layers": L - n_on, "kv_tokens": S,
           "h2d_bytes_per_target_verify": h2d, "target_verify_ms": round(t_verify, 2),
           "h2d_GBps_during_verify": round(h2d / t_verify / 1e6, 2), "pure_h2d_GBps": round(pure_rate, 2),
           "pcie_link_GBps_spec": 63.0, "copy_alone_ms": round(t_copy, 2),
           "compute_alone_ms_est": round(t_compute, 2) if t_compute else None}
    if t_compute:
        hidden = t_copy + t_compute - t_verify
        out["overlap_fraction"] = round(max(0.0, min(1.0, hidden / min(t_copy, t_compute))), 3)
    return out


def predicted_for(label, world, path=None):
    """This configuration's row of the tracked scaling prediction (tools/predict_scaling.py -> profiles/r05_predicted_scaling.json):
    per-rank stage latencies measured on ONE GPU + the low / high step the model composes from them; None when absent."""
    if path is None:                                            # the newest tracked prediction (profiles/rNN_predicted_scaling.json)
        import glob
        found = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r[0-9][0-9]_predicted_scaling.json")))
        path = found[-1] if found else os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r05_predicted_scaling.json")
    try:
        cfgs = json.load(open(path))["configs"]
    except Exception:
        return None
    key = next((k for k in cfgs if k in label), None)           # "BASELINE configs[1] shapes sharded TP=8" -> "configs[1]"
    if key is None:
        return None
    c = cfgs[key]
    rank_us = next((m for m in c["measured_per_rank_us"] if m["emulated_world"] == world), None)
    pred = next((p for p in c["predictions"]["gemm_exchange"] if p["world"] == world), None)
    if rank_us is None or pred is None:
        return None
    return {"source": os.path.relpath(path, os.path.dirname(os.path.abspath(__file__))), "config": key, "world": world,
            "loop": c["loop"], "per_rank_us": {k: rank_us[k] for k in ("draft_step_us", "retrieval_verify_us", "target_verify_us")},
            "exchange_form": "gemm_exchange", "low": pred["low"], "high": pred["high"]}


def stage_latencies_lockstep(llm, args, device, timed):
    """Per-rank latency (us) of the three model calls of a step — EVERY rank runs the same forwards the same number of
    times, in the same order, after the timed region (the target and retrieval verifies contain the exchanges: a rank
    cannot run them alone), HIP events on the launch stream.  The reference prints its per-stage latencies from every
    configuration (test/offloading_TP.py:104-119); the predicted scaling table is written in exactly these quantities."""
    S_now, g = llm.kv_cache.seq_len, args.gamma
    ids = torch.full((1, g + 2), 100, dtype=torch.long, device=device)
    pos = torch.arange(S_now, S_now + g + 1, device=device).unsqueeze(0)

    def tv():
        llm.inference(input_ids=ids)
        llm.kv_cache.seq_len = S_now
    return {"draft_step_us": round(timed(lambda: llm.draft_run(ids[:, :3], gamma_offset=2), 5), 1),
            "retrieval_verify_us": round(timed(lambda: llm.retrieval_verify(ids[:, :g + 1], pos, args.temp, args.top_p), 5), 1),
            "target_verify_us": round(timed(tv, 3), 1)}


def exchange_cost_lockstep(llm, args, device, per_graph=64, replays=5):
    """What ONE exchange costs this rank on this group (us): the o_proj-shaped GEMM of the decode layer with its exchange
    (the form the engine selected: inside the GEMM, or GEMM -> staging -> exchange kernel, or GEMM + RCCL all-reduce)
    against the SAME GEMM with the single-rank residual epilogue, each as a hipGraph of ``per_graph`` back-to-back calls —
    all ranks in lock-step.  None when the engine has a single rank or the probe fails on any rank (agreed collectively)."""
    from triforce_amd import ops
    from triforce_amd.utils.graph_infer import _capture_error_mode
    if llm.world_size == 1 or device.type != "cuda":
        return None
    rows, hid, W = args.gamma + 1, llm.hidden_size, llm.weights
    out, ok = {}, True
    cpu_flags = dist.get_backend() != "nccl"

    class _PeerFailed(Exception):
        pass

    def _agree(mine):
        """MIN over the ranks of `mine`, in front of every collective phase: a rank that failed its set-up or a capture says so
        HERE, where every rank meets, instead of leaving its peers inside a barrier or an exchange it will never join (the
        collectives then no longer pair up: the probe hung, or waited out the exchange time-out and poisoned the group —
        advisor, round 5).  Raises on every rank when any rank reports a failure."""
        f = torch.tensor([1 if mine else 0], dtype=torch.int64, device="cpu" if cpu_flags else device)
        dist.all_reduce(f, dist.ReduceOp.MIN)
        if not int(f.item()):
            raise _PeerFailed("a rank failed the exchange probe")
    try:
        packed = ops.act_packed(rows)
        a0 = torch.randn(rows, W.wo[0].K, device=device).to(torch.float16) * 0.05
        x0 = torch.zeros(rows, hid, dtype=torch.float16, device=device)
        a = ops.Act.from_rows(a0) if packed else a0
        x = ops.Act.from_rows(x0) if packed else x0
        ss = ops.ss_buffer(hid, device)

        def with_exchange():
            if llm._xchg is not None:
                llm._xchg.linear_reduce(a, W.wo[0], x, ss)
            elif llm._ar is not None:
                llm._ar.reduce(ops.linear(a, W.wo[0], out=llm._ar.staging(rows, hid, packed=packed)), x, resid=x, ss_out=ss)
            else:
                part = ops.linear(a0, W.wo[0])
                dist.all_reduce(part, dist.ReduceOp.SUM)
                x0.add_(part)

        def without():
            ops.linear(a, W.wo[0], resid=x, out=x, ss_out=ss)

        def graph_us(fn):
            fn()
            torch.cuda.synchronize(device)
            side = torch.cuda.Stream(device=device)
            side.wait_stream(torch.cuda.current_stream(device))
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side, capture_error_mode=_capture_error_mode()):
                    for _ in range(per_graph):
                        fn()
            torch.cuda.current_stream(device).wait_stream(side)
            graph.replay()
            torch.cuda.synchronize(device)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(replays):
                graph.replay()
            e.record()
            torch.cuda.synchronize(device)
            return s.elapsed_time(e) / (replays * per_graph) * 1e3
        _agree(True)                                             # set-up done on every rank (else: nobody enters an exchange)
        out["gemm_with_exchange_us"] = round(graph_us(with_exchange), 2)
        _agree(True)
        out["gemm_alone_us"] = round(graph_us(without), 2)
        out["per_exchange_us"] = round(out["gemm_with_exchange_us"] - out["gemm_alone_us"], 2)
        out["shape"] = f"{rows} rows x K {W.wo[0].K} -> hidden {hid} (o_proj shard), {per_graph} calls per hipGraph x {replays} replays"
    except _PeerFailed as ex:                                  # (every rank raised at the same agreement point)
        return {"failed": str(ex)}
    except Exception as ex:                                    # a probe must never cost the bench line
        import traceback
        out = {"failed": f"{type(ex).__name__}: {ex}"[:300], "where": traceback.format_exc()[-600:]}
        try:
            _agree(False)                                      # tell the peers at THEIR next agreement point
        except _PeerFailed:
            pass
        return out
    try:
        _agree(True)                                           # the closing agreement every successful rank reaches
    except _PeerFailed as ex:
        return {"failed": str(ex)}
    return out


def multi_rank_report(per_rank, args, tcfg, world, loop, exchange_form, predicted=None):
    """The part of the N > 1 bench line the prediction is written in (pure function of gathered numbers; schema pinned by
    tests/test_host_edges_cpu.py): per-rank stage latencies, the slowest rank's figures, exchanges per forward / step, the
    measured per-exchange time, and — when profiles/r05_predicted_scaling.json has this configuration and world size — the
    predicted step next to the measured one, term by term."""
    L = tcfg.num_hidden_layers
    k = loop["inner_iterations_per_step"]
    slow = {key: max(r["stages"][key] for r in per_rank if r.get("stages")) for key in
            ("draft_step_us", "retrieval_verify_us", "target_verify_us")} if all(r.get("stages") for r in per_rank) else None
    rep = {"stage_latency_us_per_rank": [dict(rank=r["rank"], **(r.get("stages") or {})) for r in per_rank],
           "stage_latency_us_slowest_rank": slow,
           "exchange": {"form": exchange_form, "per_forward": 2 * L, "per_step": round(2 * L * (1 + k), 1),
                        "per_rank": [dict(rank=r["rank"], **(r.get("exchange") or {})) for r in per_rank]}}
    if slow is not None:
        modelled = slow["target_verify_us"] + k * slow["retrieval_verify_us"] + (k + 1) * slow["draft_step_us"]
        rep["measured_step_terms_us"] = {"target_verify": slow["target_verify_us"], "retrieval_verify": round(k * slow["retrieval_verify_us"], 1),
                                         "draft": round((k + 1) * slow["draft_step_us"], 1),
                                         "host_and_broadcasts": round(max(0.0, loop["ms_per_step"] * 1e3 - modelled), 1),
                                         "note": "exchange time is INSIDE the verify latencies here (the forwards ran on the real "
                                                 "group); the prediction lists it as its own term on top of one-GPU shard latencies"}
    if predicted is not None:
        rep["predicted"] = predicted
        if slow is not None and predicted.get("per_rank_us"):
            pr = predicted["per_rank_us"]
            rep["measured_minus_predicted_us"] = {
                key: round(slow[key] - pr[key], 1) for key in ("target_verify_us", "retrieval_verify_us", "draft_step_us") if key in pr}
    return rep


def _emit(line):
    """The one JSON line, on the process's ORIGINAL stdout (see run_tp)."""
    os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, (line + "\n").encode())


_REAL_STDOUT = None


def run_tp(args, rank, world, local):
    # stdout must carry exactly one line (rank 0's JSON).  RCCL prints a version banner on the C stdout of every rank when
    # its first communicator is created (flushed at exit), and the engine logs which exchange / graph form it chose: from
    # here on file descriptor 1 is stderr, and the JSON line is written to a saved duplicate of the original.
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)
    from bench import (_Tok, attn_roofline, baseline_config_label, cpu_baseline, forward_bytes, metric_label, resolve_weights,
                       target_config, _stage_row, _timed)
    from triforce_amd.models.aligned import parse_spec
    from triforce_amd import ops
    from triforce_amd.models.cache import StreamingLLMEvictionCache
    from triforce_amd.models.modeling_llama_68m import LlamaForCausalLM as Draft68M
    from triforce_amd.models.TP_llama import DistributedLlama, distributed_init
    from triforce_amd.utils.decoding import ReplicaCheck, TriForceRunner, _DistEngine, tp_sync_record
    from triforce_amd.utils.sampling import UniformSource

    on_gpu = torch.cuda.is_available()
    # read by DistributedLlama.init_parameters; an explicit --allreduce wins, else a TRIFORCE_ALLREDUCE the user exported
    # stays in force (it used to be overwritten with the flag's default), else auto
    if getattr(args, "allreduce", None) is None:
        args.allreduce = os.environ.get("TRIFORCE_ALLREDUCE", "auto")
        if args.allreduce not in ("auto", "oneshot", "rccl"):
            raise SystemExit(f"TRIFORCE_ALLREDUCE={args.allreduce!r}: expected auto, oneshot or rccl")
    os.environ["TRIFORCE_ALLREDUCE"] = args.allreduce
    if "RANK" not in os.environ:                                # one process, no launcher (--engine tp / --on-chip at N=1)
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    share = on_gpu and getattr(args, "share_device", False)     # every rank on cuda:0 (functional check on a 1-GPU box):
    if share:                                                   # RCCL refuses two ranks on one device, so gloo carries
        local = 0                                               # the large collectives; the decode-sized ones are the
        torch.cuda.set_device(0)                                # one-shot kernel over hipIpc mappings either way
    distributed_init("nccl" if on_gpu and not share else "gloo")    # gloo also: the CPU plumbing test (--dry-run)
    device = torch.device("cuda", local) if on_gpu else torch.device("cpu")
    if not on_gpu and not args.dry_run:
        raise SystemExit("bench.py --gpus N needs GPUs (use --dry-run for the CPU plumbing check)")
    tcfg, dcfg = target_config(args.target)
    kind, tspec, dspec, wlabel = resolve_weights(args)
    if kind == "checkpoint":
        from triforce_amd.models.llama_core import load_checkpoint_state_dict
        draft = Draft68M.from_pretrained(dspec, device_map=device)
    elif kind == "aligned":
        draft = Draft68M(dcfg, device).init_aligned(parse_spec(dspec), attn_keys=256)
    else:
        draft = Draft68M(dcfg, device).init_random(int(dspec.split(":")[1]))
    dcache = StreamingLLMEvictionCache(draft, start_size=16, recent_size=256 - 16 - args.gamma, gamma=args.gamma)
    llm = DistributedLlama(tspec, config=tcfg, local_rank=rank, world_size=world, device=device,
                           prefill=args.prefill, gen_len=args.gen_cap, temperature=args.temp, top_p=args.top_p,
                           retrieval_budget=args.budget, retrieval_chunk_size=args.chunk_size, kv_offload=True,
                           on_chip_layers=tcfg.num_hidden_layers if args.on_chip < 0 else args.on_chip,
                           draft=draft, draft_cache=dcache, gamma=args.gamma)
    llm.init_parameters(load_checkpoint_state_dict(tspec) if kind == "checkpoint" else tspec)
    if args.dry_run:                                            # launcher / rendezvous / sharding plumbing only
        shard = torch.tensor([llm.weights.H_local, llm.weights.I_local, rank], dtype=torch.int64, device=device)
        got = [torch.zeros_like(shard) for _ in range(world)]
        dist.all_gather(got, shard)
        dist.barrier()
        if rank == 0:
            _emit(json.dumps({"metric": metric_label(args.target, args.prefill), "dry_run": True,
                              "n_gpus": world, "world_size_observed": dist.get_world_size(),
                              "backend": dist.get_backend(), "shards": [g.tolist() for g in got],
                              "config": {"workload": f"{tcfg._name_or_path} TP={world}", "weights": wlabel}}))
        dist.destroy_process_group()
        return
    if not args.no_graphs:
        llm.initialize_graphs(args.gamma)
    gen = torch.Generator().manual_seed(args.seed)
    input_ids = torch.randint(3, tcfg.vocab_size, (1, args.prefill), generator=gen).to(device)

    ge = _DistEngine(llm)
    run = TriForceRunner(_Tok(), ge, args.gamma, top_k=-1, top_p=args.top_p, temperature=args.temp,
                         rng=UniformSource(device, seed=args.seed), inclusive_accept=True, sync_record=tp_sync_record(llm))
    run.health = ge.health                                      # a timed-out exchange raises instead of emitting tokens
    # TRIFORCE_TP_REPLICATED_DECISIONS=1: no record broadcasts, the ranks' streams compared by digest (utils/decoding.ReplicaCheck)
    replicas = ReplicaCheck(device) if run.sync_record is None else None
    t0 = time.time()
    llm.reset()
    if args.prefill_mode == "real":
        llm.prefill(input_ids=input_ids[:, :-1])
    else:
        llm.kv_cache.normal_(seq_len=args.prefill - 1)
    logits = llm.build_retrieval_cache(input_ids=input_ids[:, -1:])
    cal = None
    if kind == "aligned":                                       # same calibration on every rank (replicated lm_head)
        from triforce_amd.models.aligned import calibrate_llm
        cal = calibrate_llm(llm, args.gamma, args.temp, args.top_p)
    run.start(logits)
    llm.draft_run(input_ids=input_ids)
    torch.cuda.synchronize()
    t_prefill = time.time() - t0

    for _ in range(args.warmup):
        run.step()
    n0, acc0, dr0 = run.n, run.accepted_count, run.draft_count
    inject = int(os.environ.get("TRIFORCE_BENCH_INJECT_AR_ERROR", "0"))
    if inject and getattr(llm, "_ar", None) is not None:        # fault injection (tests): as if a peer had timed out
        torch.cuda.synchronize()
        llm._ar.inject_error(inject)
        if getattr(llm, "_xchg", None) is not None:             # the decode layer's exchanges live in the GEMMs then
            llm._xchg.inject_error(inject)
    if rank == 0:
        ops.ATTN_TIMER = []                                     # rank 0 samples its attention launches (HIP events)
    # every --roofline-every-th target verify runs eagerly on EVERY rank (same exchanges, same order as the captured
    # forward) so that rank 0 can bracket attention launches with HIP events inside the timed region, like bench.py
    graphed_target = bool(getattr(llm, "_target_caps", None))
    # (world > 1: every 2N-th step — a rank's eager forward is ~10 short kernels + 2 exchanges per layer, closer to the
    # host's launch rate than the one-GPU forward, so it is sampled half as often)
    run.eager_every = args.roofline_every * (2 if world > 1 else 1) if (graphed_target and args.roofline_every > 0) else 0
    in0 = run.inner_iters
    dist.barrier()
    torch.cuda.synchronize()
    t1 = time.time()
    failure = None
    try:
        for _ in range(args.steps):
            run.step()
            if replicas is not None:
                replicas(run)                                   # (inside the timed region: it is part of that loop's cost)
        if replicas is not None:
            replicas(run, force=True)
    except RuntimeError as ex:                                  # exchange timeout (run.health) / ranks off the common stream
        failure = f"{type(ex).__name__}: {ex}"
    torch.cuda.synchronize()
    if failure is None:
        dist.barrier()
    t2 = time.time()
    run.eager_every = 0
    timer, ops.ATTN_TIMER = ops.ATTN_TIMER, None
    ar_err = llm._ar.error() if getattr(llm, "_ar", None) is not None else 0
    if getattr(llm, "_xchg", None) is not None:
        ar_err = ar_err or llm._xchg.error()
    if failure is None and ar_err:
        failure = f"one-shot all-reduce error word {ar_err} after the timed region"
    want_form = getattr(args, "require_graph_form", None)
    if failure is None and want_form and getattr(llm, "graph_form", "eager") != want_form:
        failure = f"graph_form {getattr(llm, 'graph_form', 'eager')!r} != required {want_form!r}"
    if failure is None and getattr(args, "allreduce", "auto") == "oneshot" and world > 1 and getattr(llm, "_ar", None) is None:
        failure = "--allreduce oneshot but the engine is on RCCL"
    if failure is not None:                                     # fail LOUDLY: a JSON line that says so, and rc != 0
        _emit(json.dumps({"metric": metric_label(args.target, args.prefill), "value": None,
                          "failed": failure, "rank": rank, "n_gpus": world, "allreduce_error": int(ar_err),
                          "world_size_observed": dist.get_world_size(), "graph_form": getattr(llm, "graph_form", "eager"),
                          "decode_allreduce": "oneshot" if getattr(llm, "_ar", None) is not None else "rccl"}))
        os._exit(3)
    offload = offload_report(llm, args, tcfg, world, device) if llm.on_chip_layers < tcfg.num_hidden_layers else None
    elapsed = torch.tensor([t2 - t1], dtype=torch.float64, device=device)
    dist.all_reduce(elapsed, dist.ReduceOp.MAX)                 # slowest rank defines the job time
    seconds = float(elapsed.item())
    tokens = run.n - n0
    accepted, drafted = run.accepted_count - acc0, run.draft_count - dr0
    # ---- what the scaling prediction is written in, measured on THIS group by every rank in lock-step ----
    my_stages = my_exchange = None
    if offload is None and on_gpu:
        with torch.inference_mode():                             # (the engine's buffers are inference tensors)
            my_stages = stage_latencies_lockstep(llm, args, device, _timed)
            my_exchange = exchange_cost_lockstep(llm, args, device)
    per_rank = [None] * world
    dist.all_gather_object(per_rank, {"rank": rank, "stages": my_stages, "exchange": my_exchange})
    if rank == 0:
        # per-rank roofline: this rank's heads only (H / world), against ONE GPU's HBM peak
        roof = attn_roofline(timer, args.budget + args.gamma + 1, tcfg.num_attention_heads // world, tcfg.head_dim)
        if roof is not None:
            roof["kernel"] = ("attn_split_q2_kernel<128>" if args.gamma + 2 > 16 else "attn_split_kernel<128,1>") + \
                f" via tf_attn_decode[_fused], {tcfg.num_attention_heads // world} heads of this rank"
        inner_per_step = (run.inner_iters - in0) / max(args.steps, 1)
        label = baseline_config_label(args.target, args.prefill, args.budget, args.gamma, args.on_chip,
                                      tcfg.num_hidden_layers, world)
        # stage rooflines of THIS rank's shard (graph replays, HIP events): measured by ALL ranks in lock-step below
        stages, stage_rows = my_stages, None
        if stages is not None:
            S_now, g = llm.kv_cache.seq_len, args.gamma
            rvb, wl, rkv = forward_bytes(tcfg, args.budget + g + 1, world)
            tvb, _, tkv = forward_bytes(tcfg, S_now + g + 2, world)
            how = "HIP events around the engine's forwards (hipGraph replays where captured), all ranks in lock-step"
            stage_rows = [_stage_row("retrieval_verify forward (this rank's shard)", rvb, stages["retrieval_verify_us"], how),
                          _stage_row("target_verify forward (this rank's shard)", tvb, stages["target_verify_us"], how)]
        cpu = None
        if not args.no_cpu_baseline:
            try:
                cpu = cpu_baseline(args, tokens / args.steps, inner_per_step)
            except Exception as ex:
                cpu = {"value": None, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
                       "sample": f"failed: {type(ex).__name__}: {ex}"}
        _emit(json.dumps({
            "metric": metric_label(args.target, args.prefill),
            "value": round(tokens / seconds, 3), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(seconds / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"{label}, tensor-parallel engine: {tcfg._name_or_path} TriForce decode, "
                                   f"prefill {args.prefill}, budget {args.budget}, chunk {args.chunk_size}, gamma "
                                   f"{args.gamma}, T={args.temp}, top_p={args.top_p}, TP={world} "
                                   + ("with ALL RANKS ON ONE DEVICE (functional check, not a scaling point), "
                                      if share else "over RCCL/xGMI, ")
                                   + ("KV resident in HBM" if offload is None else
                                      f"KV of {offload['offloaded_layers']} layers in pinned host memory (on_chip "
                                      f"{offload['on_chip_layers']})"),
                       "parallelism": f"tp{world}", "world_size_observed": dist.get_world_size(),
                       "prefill_mode": args.prefill_mode, "weights": wlabel, "weights_kind": kind},
            "value_note": ("configured-acceptance scenario: the synthetic weights SET both acceptance rates (models/aligned.py); "
                           "tokens/s and avg_accepted_len follow from that dial" if kind == "aligned" else None),
            "aligned_calibration": cal, "offload": offload,
            "avg_accepted_len": round(accepted / max(drafted, 1) * args.gamma, 4),
            "acceptance_rate": round(accepted / max(drafted, 1), 4), "tokens": tokens,
            "tokens_per_step": round(tokens / args.steps, 3), "prefill_seconds": round(t_prefill, 2),
            "kv_seq_len": llm.kv_cache.seq_len, "graph_form": getattr(llm, "graph_form", "eager"),
            "decode_allreduce": (("one-shot peer reads inside the o_proj / down_proj GEMMs (tf_skinny_gemm_xchg)"
                                  if getattr(llm, "_xchg", None) is not None else
                                  "one-shot peer reads (tf_allreduce_oneshot_alt, alternating staging halves)"
                                  if getattr(getattr(llm, "_ar", None), "alternate", False) else
                                  "one-shot peer reads (tf_allreduce_oneshot)")) if getattr(llm, "_ar", None) is not None
            else ("rccl" if world > 1 else "none (one rank)"),
            "decisions": ("replicated on every rank (same uniform stream, bit-identical exchanges): no record broadcasts, "
                          f"{replicas.checks} stream-digest checks across the ranks in the timed region" if replicas is not None else
                          "rank 0's records broadcast (2 per inner iteration + 1 per outer step), blocking reads"),
            "ranks_share_one_device": bool(share), "allreduce_error": int(ar_err),
            "allreduce_requested": getattr(args, "allreduce", "auto"),
            "allreduce_note": getattr(llm, "allreduce_note", "") or None,
            "inner_iterations_per_step": round(inner_per_step, 3), "stage_latency_us": stages,
            # what a step costs beyond its model calls (bench.py's definition: step - [target verify + k x retrieval verify +
            # (k + 1) x draft step]), and which launch structure the loop ran
            "step_overhead_us": (round(max(0.0, seconds / args.steps * 1e6 - (stages["target_verify_us"] + inner_per_step *
                                 (stages["retrieval_verify_us"] + stages["draft_step_us"]) + stages["draft_step_us"])), 1)
                                 if stages else None),
            "loop_structure": ("one hipGraph per inner iteration (draft step, draw, retrieval verify with its exchanges, accept "
                               "test), records through the pinned mailbox, uniforms behind the device cursor"
                               if run.inner is not None else
                               "draft replay + draw + retrieval-verify replay + accept per inner iteration, device records"),
            "multi_rank": multi_rank_report(
                per_rank, args, tcfg, world,
                {"inner_iterations_per_step": inner_per_step, "ms_per_step": seconds / args.steps * 1e3},
                ("inside the o_proj / down_proj GEMMs (tf_skinny_gemm_xchg, " + getattr(llm, "xchg_form", "fence-free") + ")"
                 if getattr(llm, "_xchg", None) is not None else
                 "GEMM -> staging -> tf_allreduce_oneshot" if getattr(llm, "_ar", None) is not None else
                 ("GEMM -> RCCL all-reduce" if world > 1 else "none (one rank)")),
                predicted_for(label, world)),
            "roofline": roof, "roofline_stages": stage_rows,
            "roofline_note": None if roof else "no eager target verify was sampled (--roofline-every 0, or no target "
            "graph): no per-launch HIP events on this run", "cpu_baseline": cpu}))
    dist.barrier()
    dist.destroy_process_group()
