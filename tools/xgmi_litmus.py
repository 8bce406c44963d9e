"""Cross-device message-passing litmus for the one-shot exchange (csrc/allreduce.hip) — the memory-model assumptions of
DESIGN section 12.5 that only a multi-GPU node can confirm, tested directly and reported as a FAILURE COUNT:

  1. fine-grained peer-mapped loads are never served stale after the reader's system-scope acquire;
  2. the plain stores of the PRECEDING kernel (the "producer GEMM": here tf_ar_litmus_stage, an ordinary kernel writing a
     pattern that changes every iteration) are visible to the peers once the exchange kernel's READY flag is — also when
     both are nodes of a hipGraph;
  3. a lane's system-scope flag store is delivered after its earlier system-scope traffic (READY after staging, DONE after
     the last remote load).

Every iteration: stage (plain stores) -> the product's exchange kernel (tf_allreduce_oneshot[_alt]: READY / reduce / DONE)
-> check (every element of the sum against the value the SAME iteration's patterns give; mismatches added to a device
counter).  No host synchronisation between iterations; every few hundred iterations one rank is delayed so the others run
ahead as far as the protocol lets them.  Phase A: eager launches; phase B: the same loop captured as a hipGraph and
replayed.  One JSON line on rank 0; exit code 1 on any mismatch, time-out or sticky error.

    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/xgmi_litmus.py
    ... tools/xgmi_litmus.py --share-device        all ranks on cuda:0 over gloo (dry run on a one-GPU box)
(the reference's exchange: dist.all_reduce at models/tensor_op.py:179,326,359)"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--share-device", action="store_true")
    ap.add_argument("--iters", type=int, default=1_000_000, help="iterations of the graph phase (rounded to whole replays)")
    ap.add_argument("--eager-iters", type=int, default=20_000)
    ap.add_argument("--rows", type=int, default=18)
    ap.add_argument("--hidden", type=int, default=5120)
    ap.add_argument("--per-graph", type=int, default=200, help="iterations captured per hipGraph (even)")
    ap.add_argument("--alternate", action="store_true", help="the alternating-halves form (no DONE handshake)")
    ap.add_argument("--budget-s", type=float, default=50.0, help="stop the graph phase after this many seconds")
    ap.add_argument("--xchg", action="store_true",
                    help="litmus of the FUSED exchange instead (tf_skinny_gemm_xchg: GEMM + per-panel exchange in one launch, the "
                         "engine's default at world > 1): both forms — fence-free, then fenced — through "
                         "utils.oneshot_ar.GemmExchange.litmus, --iters iterations each; the engine runs the same at start-up")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", rank))
    dev = torch.device("cuda", 0 if args.share_device else local)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo" if args.share_device else "nccl")
    from triforce_amd import hip
    from triforce_amd.utils.oneshot_ar import OneShotAllReduce
    L = hip.lib()
    if args.xchg:
        from triforce_amd.utils.oneshot_ar import GemmExchange
        xc = GemmExchange(rank, world, dev, 32 * args.hidden)
        res = {"world": world, "share_device": args.share_device, "kernel": "tf_skinny_gemm_xchg", "forms": []}
        fails = 0
        for fenced in (False, True):
            dist.barrier()
            xc.reset()
            dist.barrier()
            xc.set_fenced(fenced)
            t0 = time.time()
            r = xc.litmus(iters=min(args.iters, 200_000), rows=min(args.rows, 32), per_graph=args.per_graph)
            r["seconds"] = round(time.time() - t0, 2)
            everyone = [None] * world
            dist.all_gather_object(everyone, {"rank": rank, "mismatched_elements": r["mismatched_elements"], "error_word": r["error_word"]})
            r["per_rank"] = everyone
            r["failures"] = sum(e["mismatched_elements"] + (1 if e["error_word"] else 0) for e in everyone)
            fails += r["failures"]
            res["forms"].append(r)
        xc.set_fenced(False)
        res["failures"], res["litmus_ok"] = fails, fails == 0
        if rank == 0:
            print(json.dumps(res), flush=True)
        xc.close()
        dist.destroy_process_group()
        sys.exit(0 if fails == 0 else 1)
    n = args.rows * args.hidden
    ar = OneShotAllReduce(rank, world, dev, 32 * args.hidden, alternate=args.alternate)
    it_dev = torch.zeros(1, dtype=torch.int32, device=dev)
    bad = torch.zeros(1, dtype=torch.int64, device=dev)
    out = torch.zeros(args.rows, args.hidden, dtype=torch.float16, device=dev)

    def vp(t):
        return ctypes.c_void_p(t.data_ptr())

    def stream():
        return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def one():
        st = ar.staging(args.rows, args.hidden)
        hip.check(L.tf_ar_litmus_stage(vp(st), n, rank, vp(it_dev), stream()), "tf_ar_litmus_stage")
        ar.reduce(st, out)
        hip.check(L.tf_ar_litmus_check(vp(out), n, world, vp(it_dev), vp(bad), stream()), "tf_ar_litmus_check")

    res = {"world": world, "share_device": args.share_device, "form": "alternating halves" if args.alternate else "READY/DONE",
           "elements_per_exchange": n}
    dist.barrier()
    # ---- phase A: eager launches, one rank delayed now and then -------------------------------------------------------
    t0 = time.time()
    for i in range(args.eager_iters):
        if i % 211 == 0 and (i // 211) % world == rank:
            torch.cuda._sleep(200_000)                       # ~0.1 ms: the peers run ahead as far as the protocol allows
        one()
    torch.cuda.synchronize(dev)
    res["eager"] = {"iterations": args.eager_iters, "seconds": round(time.time() - t0, 2),
                    "mismatched_elements": int(bad.item()), "error_word": ar.error_device()}
    dist.barrier()
    # ---- phase B: the same loop as a hipGraph ------------------------------------------------------------------------
    per = max(2, args.per_graph - args.per_graph % 2)
    bad0 = int(bad.item())
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
            for _ in range(per):
                one()
    torch.cuda.current_stream(dev).wait_stream(side)
    replays = max(1, args.iters // per)
    t0 = time.time()
    done = 0
    # all ranks must issue the same number of replays: the count is fixed up front and the time budget is checked on
    # rank 0's clock in blocks, the verdict broadcast
    block = max(1, replays // 50)
    while done < replays:
        nb = min(block, replays - done)
        for j in range(nb):
            if (done + j) % 17 == 0 and ((done + j) // 17) % world == rank:
                torch.cuda._sleep(200_000)
            graph.replay()
        done += nb
        torch.cuda.synchronize(dev)
        stop = torch.tensor([1 if (time.time() - t0) > args.budget_s else 0], dtype=torch.int64,
                            device="cpu" if args.share_device else dev)       # gloo: host tensors; RCCL: device tensors
        dist.broadcast(stop, 0)
        if int(stop.item()):
            break
    torch.cuda.synchronize(dev)
    res["graph"] = {"iterations": done * per, "iterations_per_graph": per, "seconds": round(time.time() - t0, 2),
                    "mismatched_elements": int(bad.item()) - bad0, "error_word": ar.error_device()}
    fails = int(bad.item()) + (1 if ar.error_device() else 0)
    everyone = [None] * world
    dist.all_gather_object(everyone, {"rank": rank, "mismatched_elements": int(bad.item()), "error_word": ar.error_device()})
    res["per_rank"] = everyone
    res["failures"] = sum(e["mismatched_elements"] + (1 if e["error_word"] else 0) for e in everyone)
    res["litmus_ok"] = res["failures"] == 0
    if rank == 0:
        print(json.dumps(res), flush=True)
    ar.close()
    dist.destroy_process_group()
    sys.exit(0 if fails == 0 and res["litmus_ok"] else 1)


if __name__ == "__main__":
    main()
