# torchrun --nproc_per_node=2 test/offloading_seqouia.py --budget 12288 --prefill 130048 --target llama-7B-128K --on_chip 9 --seed 1
"""Entry point #3 — TriForce with a Sequoia static tree on the retrieval cache (tensor-parallel + KV offloading),
same flags / printed metrics as the reference's test/offloading_seqouia.py (:42-61 flags, :96-210 flow; the
reference's spelling of the file name is kept).  One process per GPU (torchrun), RCCL.

Offline additions: --weights random:<seed> | <local HF dir>, --tokenizer, --dataset synthetic (default);
--tree_size is a node count (built with triforce_amd/utils/tree.py and cached under tree/<n>.json) or a path to a
grow map in the reference's tree/<n>.pt format; --on_chip may equal the layer count."""
import argparse
import os
import sys
import time

root_dir = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.append(root_dir)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from triforce_amd.data.dataset import get_dataset, load_tokenizer  # noqa: E402
from triforce_amd.models import zoo  # noqa: E402
from triforce_amd.models.llama_core import load_checkpoint_state_dict  # noqa: E402
from triforce_amd.models.TP_llama_tree import DistributedLlama, distributed_init  # noqa: E402
from triforce_amd.utils.decoding import Baseline_Dist  # noqa: E402
from triforce_amd.utils.misc import colored  # noqa: E402
from triforce_amd.utils.SpecTree_TP import SpecTree  # noqa: E402
from triforce_amd.utils.tree import load_grow_map  # noqa: E402

local_rank, world_size = distributed_init()
device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", local_rank)))


def parse_arguments():
    parser = argparse.ArgumentParser(description="args for offloading_seqouia.py")
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
    parser.add_argument("--tree_size", type=str, default="512")
    parser.add_argument("--weights", type=str, default="random:1", help="random:<seed> or a local HF checkpoint dir")
    parser.add_argument("--tokenizer", type=str, default="none")
    parser.add_argument("--no_graphs", action="store_true", help="grow the tree eagerly like the reference")
    return parser.parse_args()


args = parse_arguments()
torch.manual_seed(args.seed)
prefill, gen_len, temperature, top_p, retrieval_budget = args.prefill, args.gen_len, args.temp, args.top_p, args.budget
if args.target not in zoo.CONFIGS:
    raise NotImplementedError
tcfg = zoo.config(args.target)
grow_map = load_grow_map(args.tree_size)
tree_size = grow_map["size"]
tokenizer = load_tokenizer(args.tokenizer, tcfg.vocab_size)
tokenized_prompts = get_dataset(dataset_name=args.dataset, tokenizer=tokenizer, datalen=prefill, vocab_size=tcfg.vocab_size)
input_ids = tokenized_prompts[0][:, :prefill].to(device)


def load_target_weights(llm):
    for rank in range(world_size):                        # rank by rank, like the reference (:100-105)
        if local_rank == rank:
            if args.weights.startswith("random"):
                llm.init_parameters(args.weights)
            else:
                llm.init_parameters(load_checkpoint_state_dict(args.weights))
        dist.barrier()


if args.baseline:
    llm = DistributedLlama(model_name_or_path=args.weights, config=tcfg, local_rank=local_rank, world_size=world_size,
                           prefill=prefill, gen_len=gen_len, temperature=temperature, top_p=top_p, flash_attn=True,
                           retrieval_budget=0, kv_offload=True, on_chip_layers=args.on_chip, tree_size=0, device=device)
    load_target_weights(llm)
    baseline_latency, gen_tokens = Baseline_Dist(tokenizer, llm, input_ids, max_len=gen_len, temperature=temperature,
                                                 top_p=top_p, local_rank=local_rank)
    baseline_latency = baseline_latency / 1000
    if local_rank == 0:
        print(colored(f"\n[Autoregressive] average latency: {baseline_latency} s", "red"))
    dist.barrier()
else:
    llm = DistributedLlama(model_name_or_path=args.weights, config=tcfg, local_rank=local_rank, world_size=world_size,
                           prefill=prefill, gen_len=gen_len, temperature=temperature, top_p=top_p, flash_attn=True,
                           retrieval_budget=retrieval_budget, kv_offload=True, on_chip_layers=args.on_chip,
                           tree_size=tree_size, device=device)
    load_target_weights(llm)
    spectree = SpecTree(engine=llm, temperature=temperature, top_p=top_p, max_length=prefill + gen_len,
                        grow_map=grow_map, tokenizer=tokenizer, vocab_size=llm.config.vocab_size)
    if not args.no_graphs:                                # the whole tree growth as one hipGraph (single rank)
        spectree.capture_grow_graph()
    all_latency, all_acc_list = [], []
    for prompt in tokenized_prompts:
        prompt = prompt[0, :prefill].to(llm.device)
        with torch.inference_mode():
            n = 0
            next_token = spectree.prefill(prefix=prompt)
            generated_ids = next_token[0].tolist()
            acc_count_list = []
            torch.cuda.synchronize()
            time1 = time.time()
            while n < gen_len:                            # :160-185
                spectree.construct_grow_map(next_token=next_token)
                next_token, acc_count, print_tokens = spectree.verify()
                if next_token is None:
                    break
                generated_ids.extend(print_tokens[1:].tolist())
                next_token = next_token.unsqueeze(0)
                n += acc_count
                acc_count_list.append(acc_count)
            if n < 64 and gen_len >= 64:
                continue
            torch.cuda.synchronize()
            time2 = time.time()
            method_latency = (time2 - time1) / max(n, 1)
            dist.barrier()
            if local_rank == 0:
                if args.verbose:
                    print(tokenizer.decode(generated_ids, skip_special_tokens=True))
                print(f"[Avg Accepted Tokens]: {np.array(acc_count_list).mean()}")
                print(colored(f"[TriForce] average latency: {method_latency} s ({n})", "red"))
            all_latency.append(method_latency)
            all_acc_list.append(np.array(acc_count_list).mean())
    if local_rank == 0 and all_latency:
        print(f"[Overall Latency]: {np.array(all_latency).mean()}")
        print(f"[Overall Avg Accepted Tokens]: {np.array(all_acc_list).mean()}")
    dist.destroy_process_group()
