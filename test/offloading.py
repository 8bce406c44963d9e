# python test/offloading.py --prefill 130048 --budget 8192 --chunk_size 8 --gamma 6
"""Entry point — single-GPU TriForce with the whole KV cache offloaded to pinned host memory, same flags and
printed metrics as the reference's test/offloading.py (:21-40 flags, :85-128 flow).  The KV of every layer is
streamed host->device per target forward, overlapped with compute on a copy stream (the reference copies
synchronously).  Offline additions as in test/on_chip.py (--weights, --tokenizer, --dataset synthetic)."""
import argparse
import os
import sys

root_dir = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.append(root_dir)

import torch  # noqa: E402

from triforce_amd.data.dataset import get_dataset, load_tokenizer  # noqa: E402
from triforce_amd.models import zoo  # noqa: E402
from triforce_amd.models.cache import OffloadingFlashSimpleCache, RetrievalCache, StreamingLLMEvictionCache  # noqa: E402
from triforce_amd.models.modeling_llama import LlamaForCausalLM  # noqa: E402
from triforce_amd.models.modeling_llama_68m import LlamaForCausalLM as LlamaForCausalLM_68M  # noqa: E402
from triforce_amd.utils.decoding import TriForce  # noqa: E402
from triforce_amd.utils.graph_infer import GraphInferenceEngine  # noqa: E402
from triforce_amd.utils.misc import colored, print_config  # noqa: E402


def parse_arguments():
    parser = argparse.ArgumentParser(description="args for offloading.py")
    parser.add_argument("--target", type=str, default="llama-7B-128K", help="target model")
    parser.add_argument("--draft", type=str, default="llama-68M", help="draft model")
    parser.add_argument("--verbose", action="store_true", help="verbose")
    parser.add_argument("--prefill", type=int, default=32768, help="prefill length")
    parser.add_argument("--gen_len", type=int, default=256, help="generation length")
    parser.add_argument("--gamma", type=int, default=6, help="gamma")
    parser.add_argument("--dataset", type=str, default="synthetic", help="dataset")
    parser.add_argument("--temp", type=float, default=0.6, help="temperature")
    parser.add_argument("--top_p", type=float, default=0.9, help="top p")
    parser.add_argument("--budget", type=int, default=8192, help="budget")
    parser.add_argument("--draft_cache_budget", type=int, default=256, help="draft cache budget")
    parser.add_argument("--chunk_size", type=int, default=8, help="chunk size")
    parser.add_argument("--weights", type=str, default="random:1")
    parser.add_argument("--draft-weights", type=str, default="random:2")
    parser.add_argument("--tokenizer", type=str, default="none")
    return parser.parse_args()


if __name__ == "__main__":
    args = parse_arguments()
    device = "cuda:0"
    if args.target not in zoo.CONFIGS:
        raise NotImplementedError
    target = LlamaForCausalLM.from_pretrained(args.weights, torch_dtype=torch.float16, device_map=device,
                                              config=zoo.config(args.target) if args.weights.startswith("random") else None).eval()
    draft = LlamaForCausalLM_68M.from_pretrained(args.draft_weights, torch_dtype=torch.float16, device_map=device,
                                                 config=zoo.config("llama-68M") if args.draft_weights.startswith("random") else None).eval()
    tokenizer = load_tokenizer(args.tokenizer, target.config.vocab_size)
    tokenized_prompts = get_dataset(dataset_name=args.dataset, tokenizer=tokenizer, datalen=args.prefill,
                                    vocab_size=target.config.vocab_size)
    top_k, top_p, temperature = -1, args.top_p, args.temp
    prefill, gen_len, gamma, verbose = args.prefill, args.gen_len, args.gamma, args.verbose
    chunk_size, max_budget = args.chunk_size, args.budget
    print_config(draft, target, prefill, gen_len, gamma, top_k, top_p, temperature, file_path=None,
                 method="TriForce (Offloading)", spec_args={"budget": args.budget, "chunk_size": chunk_size},
                 dataset=args.dataset)

    recent_size = args.draft_cache_budget - 16 - gamma
    cache = OffloadingFlashSimpleCache(target, prefill + gen_len + 32)
    cache.set_tail(prefill, gen_len + 32)
    graph_cache = RetrievalCache(target, max_budget=max_budget, prefill=prefill, gamma=gamma, chunk_size=chunk_size)
    draft_cache = StreamingLLMEvictionCache(draft, start_size=16, recent_size=recent_size, gamma=gamma)
    graph_engine = GraphInferenceEngine(target, cache, graph_cache, draft, draft_cache)
    graph_engine.initialize_cuda_graph(gamma, probs=True, temperature=temperature, top_p=top_p)
    cache.print_status()
    graph_cache.print_status()
    draft_cache.print_status()
    print(colored(f"tokenized_prompts length: {len(tokenized_prompts)}", "green"))

    all_acceptance_rate, all_speed = [], []
    for p in tokenized_prompts:
        acceptance_rate, speed = TriForce(tokenizer, graph_engine, p.to(target.device)[:, :prefill], gamma=gamma,
                                          max_len=gen_len, top_k=top_k, top_p=top_p, temperature=temperature,
                                          verbose=verbose, file_path=None, dataset=args.dataset,
                                          spec_args={"budget": args.budget, "draft": args.draft, "chunk_size": chunk_size,
                                                     "gamma": gamma, "temperature": temperature, "top_p": top_p})
        all_acceptance_rate.append(acceptance_rate)
        all_speed.append(speed)
    method_latency = 1000 / (sum(all_speed) / len(all_speed))
    print(colored(f"average acceptance rate (NOT per token): {sum(all_acceptance_rate) / len(all_acceptance_rate)}", "red"))
    print(colored(f"[TriForce] average latency: {method_latency} ms", "red"))
