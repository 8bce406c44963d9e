# torchrun --nproc_per_node=2 test/offloading_TP.py --budget 12288 --prefill 130048 --target llama-7B-128K --on_chip 9 --gamma 16
"""Entry point #2 — tensor-parallel + KV-offloading TriForce benchmark with the flags and printed metrics of the
reference's test/offloading_TP.py.  One process per GPU (torchrun), RCCL.  Flags: triforce_amd/utils/cli.py
(reference flags + --weights / --draft-weights / --tokenizer / --no_graphs / synthetic data); --on_chip may equal the
layer count — everything resident, the natural setting with 288 GB of HBM per GPU."""
import os
import sys

sys.path.append(os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from triforce_amd.models.cache import StreamingLLMEvictionCache  # noqa: E402
from triforce_amd.models.modeling_llama_68m import LlamaForCausalLM as LlamaForCausalLM_68M  # noqa: E402
from triforce_amd.models.TP_llama import DistributedLlama, distributed_init  # noqa: E402
from triforce_amd.utils import cli  # noqa: E402
from triforce_amd.utils.decoding import Baseline_Dist, TriForce_Dist  # noqa: E402
from triforce_amd.utils.misc import colored  # noqa: E402


def make_engine(args, tcfg, rank, world, device, **kw):
    return DistributedLlama(model_name_or_path=args.weights, config=tcfg, local_rank=rank, world_size=world,
                            prefill=args.prefill, gen_len=args.gen_len, temperature=args.temp, top_p=args.top_p,
                            flash_attn=True, kv_offload=True, on_chip_layers=args.on_chip, device=device, **kw)


def run_baseline(args, tcfg, tokenizer, prompts, rank, world, device):
    llm = make_engine(args, tcfg, rank, world, device, retrieval_budget=0)
    cli.shard_weights(llm, args.weights, rank, world)
    latency_ms, _ = Baseline_Dist(tokenizer, llm, prompts[0][:, :args.prefill].to(device), max_len=args.gen_len,
                                  temperature=args.temp, top_p=args.top_p, local_rank=rank)
    if rank == 0:
        print(colored(f"\n[Autoregressive] average latency: {latency_ms / 1000} s", "red"))
    dist.barrier()


def run_triforce(args, tcfg, tokenizer, prompts, rank, world, device):
    gamma = int(args.gamma)
    draft = cli.load_causal_lm(LlamaForCausalLM_68M, cli.draft_weights(args), "llama-68M", device)      # replicated on every rank
    draft_cache = StreamingLLMEvictionCache(draft, start_size=16, recent_size=256 - 16 - gamma, gamma=gamma)
    llm = make_engine(args, tcfg, rank, world, device, retrieval_budget=args.budget, draft=draft,
                      draft_cache=draft_cache, gamma=gamma)
    cli.shard_weights(llm, args.weights, rank, world)
    if not args.no_graphs:                                # draft steps, retrieval verify, HBM-resident target verify
        llm.initialize_graphs(gamma)
    accepted, latency = [], []
    for prompt in prompts:
        avg_tokens, seconds_per_token = TriForce_Dist(tokenizer, llm, prompt[:, :args.prefill].to(llm.device), gamma=gamma,
                                                      max_len=args.gen_len, top_k=-1, top_p=args.top_p,
                                                      temperature=args.temp, verbose=False, dataset=args.dataset)
        accepted.append(avg_tokens)
        latency.append(seconds_per_token)
        if rank == 0:
            print(colored(f"\n[TriForce] average latency: {seconds_per_token} s", "red"))
            print(colored(f"[TriForce] average accepted tokens: {avg_tokens}", "red"))
    if rank == 0:
        print(f"[Overall Latency]: {np.array(latency).mean()}")
        print(f"[Overall Avg Accepted Tokens]: {np.array(accepted).mean()}")
    dist.destroy_process_group()


def main():
    rank, world = distributed_init()
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", rank)))
    args = cli.parse("offloading_TP")
    torch.manual_seed(args.seed)
    tcfg = cli.target_config(args.target)
    tokenizer, prompts = cli.load_prompts(args, tcfg.vocab_size)
    (run_baseline if args.baseline else run_triforce)(args, tcfg, tokenizer, prompts, rank, world, device)


if __name__ == "__main__":
    main()
