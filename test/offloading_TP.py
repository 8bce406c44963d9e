# torchrun --nproc_per_node=2 test/offloading_TP.py --budget 12288 --prefill 130048 --target llama-7B-128K --on_chip 9 --gamma 16
"""Entry point #2 — tensor-parallel + KV-offloading TriForce benchmark, same flags / printed metrics as the
reference's test/offloading_TP.py (:26-44 flags, :74-122 flow).  One process per GPU (torchrun), RCCL.

Offline additions: --weights random:<seed> | <local HF dir>, --tokenizer, --dataset synthetic (default);
--on_chip may equal the layer count (everything resident: the natural setting with 288 GB of HBM)."""
import argparse
import os
import sys

root_dir = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.append(root_dir)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from triforce_amd.data.dataset import get_dataset, load_tokenizer  # noqa: E402
from triforce_amd.models import zoo  # noqa: E402
from triforce_amd.models.cache import StreamingLLMEvictionCache  # noqa: E402
from triforce_amd.models.llama_core import load_checkpoint_state_dict  # noqa: E402
from triforce_amd.models.modeling_llama_68m import LlamaForCausalLM as LlamaForCausalLM_68M  # noqa: E402
from triforce_amd.models.TP_llama import DistributedLlama, distributed_init  # noqa: E402
from triforce_amd.utils.decoding import Baseline_Dist, TriForce_Dist  # noqa: E402
from triforce_amd.utils.misc import colored  # noqa: E402

local_rank, world_size = distributed_init()
device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", local_rank)))


def parse_arguments():
    parser = argparse.ArgumentParser(description="args for offloading_TP.py")
    parser.add_argument("--target", type=str, default="lwm-128K", help="target model")
    parser.add_argument("--verbose", action="store_true", help="verbose")
    parser.add_argument("--prefill", type=int, default=130048, help="prefill length")
    parser.add_argument("--gen_len", type=int, default=256, help="generation length")
    parser.add_argument("--temp", type=float, default=0.6, help="temperature")
    parser.add_argument("--top_p", type=float, default=0.9, help="top p")
    parser.add_argument("--dataset", type=str, default="synthetic", help="dataset")
    parser.add_argument("--on_chip", type=int, default=0, help="on chip layers")
    parser.add_argument("--budget", type=int, default=12288)
    parser.add_argument("--baseline", action="store_true", help="baseline")
    parser.add_argument("--file", type=str, default="")
    parser.add_argument("--seed", type=int, default=1, help="seed")
    parser.add_argument("--gamma", type=str, default=6)
    parser.add_argument("--weights", type=str, default="random:1", help="random:<seed> or a local HF checkpoint dir")
    parser.add_argument("--draft-weights", type=str, default="random:2")
    parser.add_argument("--tokenizer", type=str, default="none")
    parser.add_argument("--no_graphs", action="store_true", help="run every forward eagerly like the reference")
    return parser.parse_args()


args = parse_arguments()
torch.manual_seed(args.seed)
prefill, gen_len, temperature, top_p, retrieval_budget = args.prefill, args.gen_len, args.temp, args.top_p, args.budget
if args.target not in zoo.CONFIGS:
    raise NotImplementedError
tcfg = zoo.config(args.target)
tokenizer = load_tokenizer(args.tokenizer, tcfg.vocab_size)
tokenized_prompts = get_dataset(dataset_name=args.dataset, tokenizer=tokenizer, datalen=prefill, vocab_size=tcfg.vocab_size)
input_ids = tokenized_prompts[0][:, :prefill].to(device)


def load_target_weights(llm):
    """Rank by rank, like the reference (:97-102): load (or draw) the weights, keep this rank's shard."""
    for rank in range(world_size):
        if local_rank == rank:
            if args.weights.startswith("random"):
                llm.init_parameters(args.weights)
            else:
                llm.init_parameters(load_checkpoint_state_dict(args.weights))
        dist.barrier()


if args.baseline:
    llm = DistributedLlama(model_name_or_path=args.weights, config=tcfg, local_rank=local_rank, world_size=world_size,
                           prefill=prefill, gen_len=gen_len, temperature=temperature, top_p=top_p, flash_attn=True,
                           retrieval_budget=0, kv_offload=True, on_chip_layers=args.on_chip, device=device)
    load_target_weights(llm)
    baseline_latency, gen_tokens = Baseline_Dist(tokenizer, llm, input_ids, max_len=gen_len, temperature=temperature,
                                                 top_p=top_p, local_rank=local_rank)
    baseline_latency = baseline_latency / 1000
    if local_rank == 0:
        print(colored(f"\n[Autoregressive] average latency: {baseline_latency} s", "red"))
    dist.barrier()
else:
    gamma = int(args.gamma)
    draft = LlamaForCausalLM_68M.from_pretrained(args.draft_weights, torch_dtype=torch.float16, device_map=device,
                                                 config=zoo.config("llama-68M") if args.draft_weights.startswith("random") else None).eval()
    draft_cache = StreamingLLMEvictionCache(draft, start_size=16, recent_size=256 - 16 - gamma, gamma=gamma)
    llm = DistributedLlama(model_name_or_path=args.weights, config=tcfg, local_rank=local_rank, world_size=world_size,
                           prefill=prefill, gen_len=gen_len, temperature=temperature, top_p=top_p, flash_attn=True,
                           retrieval_budget=retrieval_budget, kv_offload=True, on_chip_layers=args.on_chip, draft=draft,
                           draft_cache=draft_cache, gamma=gamma, device=device)
    load_target_weights(llm)
    if not args.no_graphs:                                # draft steps, retrieval verify and (HBM-resident) target verify
        llm.initialize_graphs(gamma)
    all_avg_tokens, all_latency = [], []
    for prompt in tokenized_prompts:
        prompt = prompt[:, :prefill].to(llm.device)
        avg_tokens, latency = TriForce_Dist(tokenizer, llm, prompt, gamma=gamma, max_len=gen_len, top_k=-1, top_p=top_p,
                                            temperature=temperature, verbose=False, file_path=None, dataset=args.dataset)
        all_avg_tokens.append(avg_tokens)
        all_latency.append(latency)
        if local_rank == 0:
            print(colored(f"\n[TriForce] average latency: {latency} s", "red"))
            print(colored(f"[TriForce] average accepted tokens: {avg_tokens}", "red"))
    if local_rank == 0:
        print(f"[Overall Latency]: {np.array(all_latency).mean()}")
        print(f"[Overall Avg Accepted Tokens]: {np.array(all_avg_tokens).mean()}")
    dist.destroy_process_group()
