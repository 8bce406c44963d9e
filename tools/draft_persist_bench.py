"""Draft step of the real Llama-68M shape: 13-launch chain (tf_draft_forward_68m) against the one-launch form
(tf_draft_forward_68m_persist), both replayed from hipGraphs over a warm StreamingLLM cache, HIP events on the launch stream.
With --stamps the persistent launch also writes per-workgroup wall-clock stamps (100 MHz) at the end of every role; the
summary is the median time since the first stamp of the launch, per role group and stamp index — the phase timeline of
DESIGN section 15.  One JSON line per configuration on stdout.

    python tools/draft_persist_bench.py [--rows 3] [--reps 300] [--stamps] [--out gpurun_out/draft_persist.jsonl]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from triforce_amd import hip, ops  # noqa: E402
from triforce_amd.models import zoo  # noqa: E402

DEV = "cuda:0"
GROUPS = {"qkv+gate|up (0..143)": range(0, 144), "gate|up only (144..191)": range(144, 192),
          "attention (192..203)": range(192, 204), "o_proj+down (204..251)": range(204, 252), "spare (252..255)": range(252, 256)}


def build(persist, gamma, peaked=True):
    from triforce_amd.models.cache import StreamingLLMEvictionCache
    from triforce_amd.models.config_yarn import LlamaConfig
    from triforce_amd.models.modeling_llama_68m import LlamaForCausalLM as Draft
    cfg = dict(zoo.CONFIGS["llama-68M"])
    cfg.update(vocab_size=32000, num_key_value_heads=12, rope_theta=10000.0, rope_scaling=None, hidden_act="silu")
    ops.DRAFT_PERSIST = persist
    g = torch.Generator().manual_seed(11)
    H, I, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]

    def w(*shape, s=0.02):
        return (torch.randn(*shape, generator=g) * s).to(torch.float16)
    sd = {"model.embed_tokens.weight": w(V, H), "model.norm.weight": torch.ones(H, dtype=torch.float16), "lm_head.weight": w(V, H, s=0.4 if peaked else 0.02)}
    for i in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{i}."
        for nme, shp in (("self_attn.q_proj", (H, H)), ("self_attn.k_proj", (H, H)), ("self_attn.v_proj", (H, H)),
                         ("self_attn.o_proj", (H, H)), ("mlp.gate_proj", (I, H)), ("mlp.up_proj", (I, H)), ("mlp.down_proj", (H, I))):
            sd[p + nme + ".weight"] = w(*shp)
        sd[p + "input_layernorm.weight"] = torch.ones(H, dtype=torch.float16)
        sd[p + "post_attention_layernorm.weight"] = torch.ones(H, dtype=torch.float16)
    # (peaked: logits of std ~10 — a handful of entries carry the mass, like a trained draft's rows; flat: std ~0.5)
    m = Draft.from_state_dict(LlamaConfig.from_dict(cfg), sd, DEV)
    c = StreamingLLMEvictionCache(m, start_size=16, recent_size=256 - 16 - gamma, gamma=gamma)
    gen = torch.Generator().manual_seed(1)
    for _ in range(20):                                                # a full window
        m.forward(torch.randint(3, 32000, (1, 16), generator=gen).to(DEV), c, None, -1)
        if c.seq_len + 16 > c.start_size + c.recent_size:
            break
    return m, c


def graph_of(m, c, rows, probs):
    ids = torch.randint(3, 32000, (1, rows), generator=torch.Generator().manual_seed(100 + rows)).to(DEV)
    kw = dict(probs=(0.6, 0.9)) if probs else {}
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            m.forward(ids, c, c, rows - 1, **kw)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = m.forward(ids, c, c, rows - 1, **kw)
    return g, out


def timed(g, reps):
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def stamps_summary(m, c, rows, launches=20):
    L = hip.lib()
    per = L.tf_draft_persist_stamps(None)
    buf = torch.zeros(256 * per, dtype=torch.int64, device=DEV)
    ids = torch.randint(3, 32000, (1, rows), generator=torch.Generator().manual_seed(100 + rows)).to(DEV)
    acc = {}
    L.tf_draft_persist_stamps(ops._ptr(buf))
    try:
        for it in range(launches):
            buf.zero_()
            m.forward(ids, c, c, rows - 1, probs=(0.6, 0.9))
            torch.cuda.synchronize()
            st = buf.view(256, per).cpu()
            t0 = int(st[:, 0].min())
            if it < 2:
                continue
            for name, rng in GROUPS.items():
                rows_ = st[list(rng)]
                nvalid = int((rows_[0] != 0).sum())
                rel = (rows_[:, :nvalid] - t0).float() / 100.0          # us
                acc.setdefault(name, []).append(rel.median(dim=0).values)
    finally:
        L.tf_draft_persist_stamps(None)
    return {k: [round(float(x), 2) for x in torch.stack(v).median(dim=0).values] for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, nargs="*", default=[1, 3, 7])
    ap.add_argument("--reps", type=int, default=300)
    ap.add_argument("--gamma", type=int, default=6)
    ap.add_argument("--stamps", action="store_true")
    ap.add_argument("--flat", action="store_true", help="N(0, 0.02) weights only: flat probability rows (the worst case of the top-p select)")
    ap.add_argument("--tag", default="")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    lines = []
    with torch.inference_mode():
        mc, cc = build(False, a.gamma, not a.flat)
        mp, cp = build(True, a.gamma, not a.flat)
        assert mp._persist is not None, "the one-launch form refused this device / shape"
        for rows in a.rows:
            for probs in (True, False):
                ops.DRAFT_PERSIST = False
                gc_, oc = graph_of(mc, cc, rows, probs)
                ops.DRAFT_PERSIST = True
                gp, op = graph_of(mp, cp, rows, probs)
                tc, tp = timed(gc_, a.reps), timed(gp, a.reps)
                tc2, tp2 = timed(gc_, a.reps), timed(gp, a.reps)
                line = {"what": "draft step, 68M shape, graph replay", "tag": a.tag, "rows_kind": "flat" if a.flat else "peaked", "rows": rows, "probs": probs, "kv_len": 256 + rows,
                        "chain_us": round(min(tc, tc2), 2), "persist_us": round(min(tp, tp2), 2),
                        "ratio": round(min(tp, tp2) / min(tc, tc2), 3), "bytes": 87e6,
                        "persist_frac_of_8TBps": round(87e6 / (min(tp, tp2) * 1e-6) / 8e12, 4), "error": mp._persist.error()}
                lines.append(line)
                print(json.dumps(line), flush=True)
        if a.stamps:
            ops.DRAFT_PERSIST = True
            for rows in a.rows[:2]:
                line = {"what": "persistent draft: median us since launch start at the end of each role, per workgroup group",
                        "tag": a.tag, "rows_kind": "flat" if a.flat else "peaked", "rows": rows, "timeline": stamps_summary(mp, cp, rows)}
                lines.append(line)
                print(json.dumps(line), flush=True)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "a") as f:
            for ln in lines:
                f.write(json.dumps(ln) + "\n")


if __name__ == "__main__":
    main()
