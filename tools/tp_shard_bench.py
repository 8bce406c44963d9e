"""Per-rank compute of the tensor-parallel decode step at the BASELINE configs[3] / configs[4] shard shapes, on ONE GPU:
builds rank 0's shard of a W-way DistributedLlama (heads / MLP columns / KV cache divided by W, all layers resident,
whole-forward hipGraphs) with a single-process group — every exchange step degenerates to a no-op, so the numbers are
what one rank COMPUTES per forward (logits are not meaningful).  What the driver's 8-GPU run adds on top is the exchange
time.
    python tools/tp_shard_bench.py llama-7B-128K 2 [--gamma 16 --prefill 130048 --budget 12288]
    python tools/tp_shard_bench.py llama-13B-128K 8
"""
import argparse
import json
import os
import socket
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TRIFORCE_ONESHOT_AR"] = "0"            # no peers to map in a one-process group


def timed(fn, n=4):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("target")
    ap.add_argument("world", type=int)
    ap.add_argument("--prefill", type=int, default=130048)
    ap.add_argument("--budget", type=int, default=12288)
    ap.add_argument("--gamma", type=int, default=16)
    ap.add_argument("--local-exchange", action="store_true",
                    help="give the engine a ONE-rank one-shot all-reduce (the real kernel, no peer to wait for): every "
                         "exchange step is launched like on a W-rank job and the fused decode layer (which needs it for "
                         "the sum-of-squares hand-off) is the one measured; without it exchanges are no-ops and the "
                         "layer is the un-fused one")
    ap.add_argument("--gemm-exchange", action="store_true",
                    help="with --local-exchange: o_proj / down_proj with the exchange in their epilogue (tf_skinny_gemm_xchg, "
                         "one-rank group) instead of GEMM -> staging -> exchange kernel")
    ap.add_argument("--alternate", action="store_true",
                    help="with --local-exchange: the alternating-halves form of the exchange (tf_allreduce_oneshot_alt)")
    args = ap.parse_args()
    from triforce_amd.models import zoo
    from triforce_amd.models.cache import StreamingLLMEvictionCache
    from triforce_amd.models.modeling_llama_68m import LlamaForCausalLM as Draft68M
    from triforce_amd.models.TP_llama import DistributedLlama
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl")
    dev = torch.device("cuda", 0)
    tcfg, dcfg = zoo.config(args.target), zoo.config("llama-68M")
    W, g = args.world, args.gamma
    draft = Draft68M(dcfg, dev).init_random(2)
    dcache = StreamingLLMEvictionCache(draft, start_size=16, recent_size=256 - 16 - g, gamma=g)
    llm = DistributedLlama("random:1", config=tcfg, local_rank=0, world_size=W, device=dev, prefill=args.prefill,
                           gen_len=256, retrieval_budget=args.budget, retrieval_chunk_size=8, kv_offload=True,
                           on_chip_layers=tcfg.num_hidden_layers, draft=draft, draft_cache=dcache, gamma=g)
    llm.init_parameters("random:1")
    if args.local_exchange and W > 1:
        from triforce_amd.utils.oneshot_ar import OneShotAllReduce
        llm._ar = OneShotAllReduce.local_group(1, dev, llm.ONESHOT_MAX_ROWS * llm.hidden_size,
                                               alternate=args.alternate)[0]
        if args.gemm_exchange:
            from triforce_amd.utils.oneshot_ar import GemmExchange
            llm._xchg = GemmExchange.local_group(1, dev, llm.ONESHOT_MAX_ROWS * llm.hidden_size)[0]
    llm.initialize_graphs(g)
    gen = torch.Generator(device=dev).manual_seed(3)
    for t in (llm.kv_cache.k, llm.kv_cache.v, llm.retrieval_cache.k, llm.retrieval_cache.v, dcache.k, dcache.v):
        for l in range(t.shape[0]):
            t[l].normal_(generator=gen)
    S = args.prefill
    llm.kv_cache.seq_len = S
    ids = torch.full((1, g + 2), 100, dtype=torch.long, device=dev)
    pos = torch.arange(S, S + g + 1, device=dev).unsqueeze(0)

    def tv():
        llm.inference(input_ids=ids)
        llm.kv_cache.seq_len = S

    def ar():
        llm.inference(input_ids=ids[:, :1])
        llm.kv_cache.seq_len = S
    Hl, D, L = tcfg.num_attention_heads // W, tcfg.head_dim, tcfg.num_hidden_layers
    out = {"target": args.target, "emulated_world": W, "heads_per_rank": Hl, "layers": L, "prefill": S, "budget": args.budget,
           "gamma": g, "graph_form": llm.graph_form,
           "exchange": ("inside the o_proj / down_proj GEMMs (one-rank group)" if llm._xchg is not None else
                        "one-rank one-shot kernel" + (", alternating halves" if args.alternate else ""))
           if llm._ar is not None else "no-op",
           "decode_layer": ("fused (6 launches)" if llm._xchg is not None else "fused (8 launches)")
           if llm._fused_decode(g + 1) else "un-fused (11 launches)",
           "draft_step_us": round(timed(lambda: llm.draft_run(ids[:, :3], gamma_offset=2)), 1),
           "retrieval_verify_us": round(timed(lambda: llm.retrieval_verify(ids[:, :g + 1], pos)), 1),
           "target_verify_us": round(timed(tv, 3), 1), "ar_step_eager_us": round(timed(ar, 3), 1)}
    kv_bytes = 2 * (S + g + 2) * Hl * D * 2 * L
    out["target_verify_attention_bytes_per_rank"] = kv_bytes
    out["exchanges_per_forward"] = 2 * L
    print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
