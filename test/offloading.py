# python test/offloading.py --prefill 130048 --budget 8192 --chunk_size 8 --gamma 6
"""Entry point — single-GPU TriForce with the whole KV cache in pinned host memory (flags and printed metrics of the
reference's test/offloading.py).  Every target forward streams the KV of each layer host->device on a copy stream,
overlapped with the previous layer's compute (the reference copies synchronously).  Flags: triforce_amd/utils/cli.py."""
import os
import sys

sys.path.append(os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

from triforce_amd.models.cache import OffloadingFlashSimpleCache, RetrievalCache, StreamingLLMEvictionCache  # noqa: E402
from triforce_amd.models.modeling_llama import LlamaForCausalLM  # noqa: E402
from triforce_amd.models.modeling_llama_68m import LlamaForCausalLM as LlamaForCausalLM_68M  # noqa: E402
from triforce_amd.utils import cli  # noqa: E402
from triforce_amd.utils.decoding import TriForce  # noqa: E402
from triforce_amd.utils.graph_infer import GraphInferenceEngine  # noqa: E402
from triforce_amd.utils.misc import colored, print_config  # noqa: E402

DEVICE = "cuda:0"


def main():
    args = cli.parse("offloading")
    cli.target_config(args.target)
    target = cli.load_causal_lm(LlamaForCausalLM, args.weights, args.target, DEVICE)
    draft = cli.load_causal_lm(LlamaForCausalLM_68M, cli.draft_weights(args), "llama-68M", DEVICE)
    tokenizer, prompts = cli.load_prompts(args, target.config.vocab_size)
    sampling = dict(top_k=-1, top_p=args.top_p, temperature=args.temp)
    print_config(draft, target, args.prefill, args.gen_len, args.gamma, file_path=None, method="TriForce (Offloading)",
                 spec_args={"budget": args.budget, "chunk_size": args.chunk_size}, dataset=args.dataset, **sampling)

    host_kv = OffloadingFlashSimpleCache(target, args.prefill + args.gen_len + 32)
    host_kv.set_tail(args.prefill, args.gen_len + 32)          # device mirror of the generated rows (retrieval tail)
    retrieval = RetrievalCache(target, max_budget=args.budget, prefill=args.prefill, gamma=args.gamma,
                               chunk_size=args.chunk_size)
    streaming = StreamingLLMEvictionCache(draft, start_size=16, recent_size=args.draft_cache_budget - 16 - args.gamma,
                                          gamma=args.gamma)
    engine = GraphInferenceEngine(target, host_kv, retrieval, draft, streaming)
    engine.initialize_cuda_graph(args.gamma, probs=True, temperature=args.temp, top_p=args.top_p)
    for c in (host_kv, retrieval, streaming):
        c.print_status()
    print(colored(f"tokenized_prompts length: {len(prompts)}", "green"))

    spec_args = {"budget": args.budget, "draft": args.draft, "chunk_size": args.chunk_size, "gamma": args.gamma,
                 "temperature": args.temp, "top_p": args.top_p}
    results = [TriForce(tokenizer, engine, p.to(target.device)[:, :args.prefill], gamma=args.gamma, max_len=args.gen_len,
                        verbose=args.verbose, file_path=None, dataset=args.dataset, spec_args=spec_args, **sampling)
               for p in prompts]
    print(colored(f"average acceptance rate (NOT per token): {cli.mean([acc for acc, _ in results])}", "red"))
    print(colored(f"[TriForce] average latency: {1000 / cli.mean([speed for _, speed in results])} ms", "red"))


if __name__ == "__main__":
    main()
