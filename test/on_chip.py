# python test/on_chip.py --prefill 124928 --budget 4096 --chunk_size 8 --top_p 0.9 --temp 0.6 --gamma 6
"""Entry point #1 — single-GPU on-chip TriForce benchmark with the flags and printed metrics of the reference's
test/on_chip.py: autoregressive baseline, then TriForce, latency and end-to-end speed-up.  Flags live in
triforce_amd/utils/cli.py (reference flags + --weights / --draft-weights / --tokenizer / --greedy / synthetic data)."""
import os
import sys

sys.path.append(os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

from triforce_amd.models.cache import FlashSimpleCache, RetrievalCache, StreamingLLMEvictionCache  # noqa: E402
from triforce_amd.models.modeling_llama import LlamaForCausalLM  # noqa: E402
from triforce_amd.models.modeling_llama_68m import LlamaForCausalLM as LlamaForCausalLM_68M  # noqa: E402
from triforce_amd.utils import cli  # noqa: E402
from triforce_amd.utils.decoding import Autoregressive, TriForce  # noqa: E402
from triforce_amd.utils.graph_infer import GraphInferenceEngine  # noqa: E402
from triforce_amd.utils.misc import colored, print_config  # noqa: E402

DEVICE = "cuda:0"


def build_engine(args, target, draft):
    """Caches sized as on_chip.py:76-80 (16 sink tokens, recent = draft budget - 16 - gamma) and the captured graphs."""
    full = FlashSimpleCache(target, args.prefill + args.gen_len + 16)
    retrieval = RetrievalCache(target, max_budget=args.budget, prefill=args.prefill, gamma=args.gamma,
                               chunk_size=args.chunk_size)
    streaming = StreamingLLMEvictionCache(draft, start_size=16, recent_size=args.draft_cache_budget - 16 - args.gamma,
                                          gamma=args.gamma)
    engine = GraphInferenceEngine(target, full, retrieval, draft, streaming)
    engine.initialize_cuda_graph(args.gamma, probs=True, temperature=args.temp, top_p=args.top_p)
    for c in (full, retrieval, streaming):
        c.print_status()
    return engine


def main():
    args = cli.parse("on_chip")
    cli.target_config(args.target)
    target = cli.load_causal_lm(LlamaForCausalLM, args.weights, args.target, DEVICE)
    draft = cli.load_causal_lm(LlamaForCausalLM_68M, cli.draft_weights(args), "llama-68M", DEVICE)
    tokenizer, prompts = cli.load_prompts(args, target.config.vocab_size)
    sampling = dict(top_k=-1, top_p=args.top_p, temperature=args.temp)
    print_config(draft, target, args.prefill, args.gen_len, args.gamma, file_path=args.file, method="TriForce",
                 spec_args={"budget": args.budget, "chunk_size": args.chunk_size}, dataset=args.dataset, **sampling)
    engine = build_engine(args, target, draft)
    print(colored(f"tokenized_prompts length: {len(prompts)}", "green"))

    def clip(p):
        return p.to(target.device)[:, :args.prefill]

    # ---- autoregressive baseline (one warm-up run, then the first prompt timed) ----
    Autoregressive(tokenizer, engine, clip(prompts[0]), max_len=args.gen_len, verbose=args.verbose, **sampling)
    ar_speed = [Autoregressive(tokenizer, engine, clip(p), max_len=args.gen_len, verbose=args.verbose, **sampling)
                for p in prompts[:1]]
    baseline_latency = 1000 / cli.mean(ar_speed)
    print(colored(f"[Autoregressive] average latency: {baseline_latency} ms", "red"))

    # ---- TriForce (three warm-up runs, which also leave the retrieval cache's init_graph set) ----
    spec_args = {"budget": args.budget, "draft": args.draft, "chunk_size": args.chunk_size, "gamma": args.gamma,
                 "temperature": args.temp, "top_p": args.top_p, "baseline": baseline_latency / 1000}
    run = dict(gamma=args.gamma, max_len=args.gen_len, verbose=args.verbose, dataset=args.dataset, spec_args=spec_args,
               rebuild_every=args.rebuild_every, **sampling)
    for _ in range(3):
        TriForce(tokenizer, engine, clip(prompts[0]), **run)
    results = [TriForce(tokenizer, engine, clip(p), file_path=args.file, **run) for p in prompts]
    method_latency = 1000 / cli.mean([speed for _, speed in results])
    print(colored(f"average acceptance rate (NOT per token): {cli.mean([acc for acc, _ in results])}", "red"))
    print(colored(f"[TriForce] average latency: {method_latency} ms", "red"))
    print(colored(f"[E2E Speedup]: {baseline_latency / method_latency}", "red"))


if __name__ == "__main__":
    main()
