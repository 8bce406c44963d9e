"""Shared plumbing of the four entry scripts (test/on_chip.py, test/offloading.py, test/offloading_TP.py,
test/offloading_seqouia.py): the command lines of the reference's scripts as declarative flag tables (same flag
names, types and defaults — reference test/on_chip.py:21-40, test/offloading.py:21-40, test/offloading_TP.py:26-44,
test/offloading_seqouia.py:42-61) plus the offline additions, and the model / prompt loaders they share.

Flag table rows: (name, type | "flag", default, help).
"""
import argparse

import torch

# ---- flags every script of the reference has ------------------------------------------------------------------
_COMMON = [
    ("--verbose", "flag", None, "verbose"),
    ("--gen_len", int, 256, "generation length"),
    ("--temp", float, 0.6, "temperature"),
    ("--top_p", float, 0.9, "top p"),
    ("--dataset", str, "synthetic", "dataset (synthetic | gs | 128k | one-shot | demo | lwm)"),
]
# ---- offline additions (the reference hard-codes hub ids) -------------------------------------------------------
_OFFLINE = [
    ("--weights", str, "random:1", "random:<seed>, aligned[:draft_acc[:retrieval_acc[:seed]]] (synthetic pair with set "
                                   "acceptance rates, models/aligned.py) or a local HF checkpoint dir"),
    ("--tokenizer", str, "none", "local tokenizer dir, or none"),
]
_SINGLE_GPU = [
    ("--target", str, "llama-7B-128K", "target model"),
    ("--draft", str, "llama-68M", "draft model"),
    ("--prefill", int, 32768, "prefill length"),
    ("--gamma", int, 6, "gamma"),
    ("--draft_cache_budget", int, 256, "draft cache budget"),
    ("--chunk_size", int, 8, "chunk size"),
    ("--draft-weights", str, "random:2", "random:<seed> or a local HF checkpoint dir"),
]
_TENSOR_PARALLEL = [
    ("--target", str, "lwm-128K", "target model"),
    ("--prefill", int, 130048, "prefill length"),
    ("--on_chip", int, 0, "on chip layers"),
    ("--budget", int, 12288, "retrieval budget"),
    ("--baseline", "flag", None, "baseline"),
    ("--file", str, "", "CSV log path"),
    ("--seed", int, 1, "seed"),
    ("--no_graphs", "flag", None, "run every forward eagerly like the reference"),
]
SCRIPTS = {
    "on_chip": _COMMON + _OFFLINE + _SINGLE_GPU + [
        ("--budget", int, 4096, "retrieval budget"),
        ("--greedy", "flag", None, "temperature 1.0, top_p 1e-9 (the only greedy the reference's sampler admits)"),
        ("--file", str, None, "CSV log path"),
        ("--rebuild_every", int, 0, "re-select the retrieval cache every N target verifies (0 = once per prompt)"),
    ],
    "offloading": _COMMON + _OFFLINE + _SINGLE_GPU + [("--budget", int, 8192, "retrieval budget")],
    "offloading_TP": _COMMON + _OFFLINE + _TENSOR_PARALLEL + [
        ("--gamma", str, 6, "gamma"),
        ("--draft-weights", str, "random:2", "random:<seed> or a local HF checkpoint dir"),
    ],
    "offloading_seqouia": _COMMON + _OFFLINE + _TENSOR_PARALLEL + [
        ("--tree_size", str, "512", "node count, or a grow map file (tree/<n>.pt / .json)"),
    ],
}


def parse(script, argv=None):
    ap = argparse.ArgumentParser(description=f"args for {script}.py")
    for name, typ, default, help_ in SCRIPTS[script]:
        if typ == "flag":
            ap.add_argument(name, action="store_true", help=help_)
        else:
            ap.add_argument(name, type=typ, default=default, help=help_)
    args = ap.parse_args(argv)
    if getattr(args, "greedy", False):
        args.temp, args.top_p = 1.0, 1e-9
    return args


# ---- loaders ----------------------------------------------------------------------------------------------------
def target_config(name):
    from ..models import zoo
    if name not in zoo.CONFIGS:
        raise NotImplementedError(name)
    return zoo.config(name)


def load_causal_lm(cls, weights, config_name, device):
    """``random:<seed>`` -> random init of the named architecture; anything else is a local HF checkpoint directory."""
    from ..models import zoo
    cfg = zoo.config(config_name) if weights.startswith(("random", "aligned")) else None
    return cls.from_pretrained(weights, torch_dtype=torch.float16, device_map=device, config=cfg).eval()


def draft_weights(args):
    """The 68M draft's weight spec: an aligned synthetic target brings its aligned draft (same planted table)."""
    return args.weights if args.weights.startswith("aligned") else args.draft_weights


def load_prompts(args, vocab_size):
    from ..data.dataset import get_dataset, load_tokenizer
    tokenizer = load_tokenizer(args.tokenizer, vocab_size)
    prompts = get_dataset(dataset_name=args.dataset, tokenizer=tokenizer, datalen=args.prefill, vocab_size=vocab_size)
    return tokenizer, prompts


def shard_weights(llm, weights, local_rank, world_size):
    """Rank by rank like the reference (offloading_TP.py:97-102): load or draw the weights, keep this rank's shard."""
    import torch.distributed as dist
    from ..models.llama_core import load_checkpoint_state_dict
    for rank in range(world_size):
        if local_rank == rank:
            llm.init_parameters(weights if weights.startswith(("random", "aligned"))
                                else load_checkpoint_state_dict(weights))
        dist.barrier()


def mean(xs):
    return sum(xs) / len(xs)
