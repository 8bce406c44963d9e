"""Prompt sources (role of the reference's data/dataset.py:17-74).

The reference reads PG-19 json / NarrativeQA through HF `datasets`; neither is present offline
(.MISSING_LARGE_BLOBS), so a `synthetic` source (seeded random token ids of the requested length) is the
default here.  The named sources are kept and work when the files / hub cache exist."""
import os

import torch


class NullTokenizer:
    """Stand-in when no tokenizer files are available: ids only, nothing to decode."""
    eos_token_id = 2

    def __init__(self, vocab_size=32000):
        self.vocab_size = vocab_size

    def decode(self, ids, **kw):
        return ""

    def encode(self, text, return_tensors=None):
        raise RuntimeError("NullTokenizer cannot encode text; use --dataset synthetic or pass --tokenizer <path>")


def load_tokenizer(path_or_id, vocab_size=32000):
    if path_or_id in (None, "", "none"):
        return NullTokenizer(vocab_size)
    from transformers import AutoTokenizer
    return AutoTokenizer.from_pretrained(path_or_id, use_fast=True, legacy=False)


def build_chat_input_lwm(tokenizer, message, prefill=127 * 1024):
    # single-turn chat format of LWM-Text-Chat (reference dataset.py:9-15)
    book = tokenizer.encode(message)[:prefill - 84]
    prompt = ("You are a helpful assistant. USER: Please read a part of the book below, and then give me the summary.\n"
              "[start of the book]\n" + tokenizer.decode(book, skip_special_tokens=True) +
              "\n[end of the book]\n\nNow you have read it. Please summarize it for me. First, tell me the title and "
              "the author, and then tell the story in 400 words.\n\nASSISTANT: ")
    return tokenizer.encode(prompt, return_tensors="pt")


def get_dataset(dataset_name, tokenizer=None, datalen=None, task=None, vocab_size=32000, num_prompts=1, seed=0):
    if dataset_name == "synthetic":
        g = torch.Generator().manual_seed(seed)
        return [torch.randint(3, vocab_size, (1, datalen), generator=g, dtype=torch.long) for _ in range(num_prompts)]
    if dataset_name in ("128k", "gs", "one-shot"):
        from datasets import load_dataset
        parent = "data/pg19/"
        files = [parent + n for n in os.listdir(parent)]
        ds = load_dataset("json", data_files=files, split="train")
        count = {"128k": len(ds), "gs": 20, "one-shot": 1}[dataset_name]
        return [tokenizer.encode(ds[i]["text"], return_tensors="pt") for i in range(count)]
    if dataset_name in ("demo", "lwm"):
        from datasets import load_dataset
        ds = load_dataset("narrativeqa")
        idx = [0, 50, 300, 800, 950, 1100, 2150, 2450, 2550, 2750, 3350, 3400, 3600, 3900, 4000, 4100, 4200, 4400, 4500, 4550]
        if dataset_name == "demo":
            return [build_chat_input_lwm(tokenizer, ds["train"][idx[2]]["document"]["text"][3:1024 * 500])]
        out = []
        for i in idx:
            t = build_chat_input_lwm(tokenizer, ds["train"][i]["document"]["text"][3:1024 * 500])
            if t.shape[-1] == 127 * 1024:
                out.append(t)
        return out
    raise Exception("Dataset not found")
