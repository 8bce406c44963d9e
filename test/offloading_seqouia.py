# torchrun --nproc_per_node=2 test/offloading_seqouia.py --budget 12288 --prefill 130048 --target llama-7B-128K --on_chip 9 --seed 1
"""Entry point #3 — TriForce with a Sequoia static tree on the retrieval cache (tensor-parallel + KV offloading), flags
and printed metrics of the reference's test/offloading_seqouia.py (whose file-name spelling is kept).  One process per
GPU (torchrun), RCCL.  Flags: triforce_amd/utils/cli.py; --tree_size is a node count (built by
triforce_amd/utils/tree.py, cached as tree/<n>.json) or a grow-map file in the reference's tree/<n>.pt format."""
import os
import sys
import time

sys.path.append(os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from triforce_amd.models.TP_llama_tree import DistributedLlama, distributed_init  # noqa: E402
from triforce_amd.utils import cli  # noqa: E402
from triforce_amd.utils.decoding import Baseline_Dist  # noqa: E402
from triforce_amd.utils.misc import colored  # noqa: E402
from triforce_amd.utils.SpecTree_TP import SpecTree  # noqa: E402
from triforce_amd.utils.tree import load_grow_map  # noqa: E402


def make_engine(args, tcfg, rank, world, device, budget, tree_size):
    return DistributedLlama(model_name_or_path=args.weights, config=tcfg, local_rank=rank, world_size=world,
                            prefill=args.prefill, gen_len=args.gen_len, temperature=args.temp, top_p=args.top_p,
                            flash_attn=True, retrieval_budget=budget, kv_offload=True, on_chip_layers=args.on_chip,
                            tree_size=tree_size, device=device)


def generate(spectree, prompt, gen_len):
    """One prompt through the grow / verify loop (offloading_seqouia.py:155-185).  Returns (tokens, accept counts, s)."""
    next_token = spectree.prefill(prefix=prompt)
    generated, counts, n = next_token[0].tolist(), [], 0
    torch.cuda.synchronize()
    t0 = time.time()
    while n < gen_len:
        spectree.construct_grow_map(next_token=next_token)
        next_token, acc_count, tokens = spectree.verify()
        if next_token is None:                            # eos accepted or degenerate residual
            break
        generated.extend(tokens[1:].tolist())
        next_token = next_token.unsqueeze(0)
        n += acc_count
        counts.append(acc_count)
    torch.cuda.synchronize()
    return generated, counts, n, time.time() - t0


def main():
    rank, world = distributed_init()
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", rank)))
    args = cli.parse("offloading_seqouia")
    torch.manual_seed(args.seed)
    tcfg = cli.target_config(args.target)
    tokenizer, prompts = cli.load_prompts(args, tcfg.vocab_size)
    if args.baseline:
        llm = make_engine(args, tcfg, rank, world, device, budget=0, tree_size=0)
        cli.shard_weights(llm, args.weights, rank, world)
        latency_ms, _ = Baseline_Dist(tokenizer, llm, prompts[0][:, :args.prefill].to(device), max_len=args.gen_len,
                                      temperature=args.temp, top_p=args.top_p, local_rank=rank)
        if rank == 0:
            print(colored(f"\n[Autoregressive] average latency: {latency_ms / 1000} s", "red"))
        dist.barrier()
        return

    grow_map = load_grow_map(args.tree_size)
    llm = make_engine(args, tcfg, rank, world, device, budget=args.budget, tree_size=grow_map["size"])
    cli.shard_weights(llm, args.weights, rank, world)
    spectree = SpecTree(engine=llm, temperature=args.temp, top_p=args.top_p, max_length=args.prefill + args.gen_len,
                        grow_map=grow_map, tokenizer=tokenizer, vocab_size=llm.config.vocab_size)
    if not args.no_graphs:                                # the whole tree growth as one hipGraph (single rank)
        spectree.capture_grow_graph()
    latencies, accepted = [], []
    with torch.inference_mode():
        for prompt in prompts:
            generated, counts, n, seconds = generate(spectree, prompt[0, :args.prefill].to(llm.device), args.gen_len)
            if n < 64 and args.gen_len >= 64:             # the reference drops runs that ended early (:187-188)
                continue
            dist.barrier()
            if rank == 0:
                if args.verbose:
                    print(tokenizer.decode(generated, skip_special_tokens=True))
                print(f"[Avg Accepted Tokens]: {np.array(counts).mean()}")
                print(colored(f"[TriForce] average latency: {seconds / max(n, 1)} s ({n})", "red"))
            latencies.append(seconds / max(n, 1))
            accepted.append(np.array(counts).mean())
    if rank == 0 and latencies:
        print(f"[Overall Latency]: {np.array(latencies).mean()}")
        print(f"[Overall Avg Accepted Tokens]: {np.array(accepted).mean()}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
