"""Predicted strong-scaling curve of the TriForce decode loop at W = 1 / 2 / 4 / 8 ranks, composed from what ONE GPU can
measure — so that the first multi-GPU run (the driver's SCALE_rNN.json, `bash tools/gpu_validate.sh tp N`) can be judged
line by line.  (The reference's own 2-GPU table: /root/reference/index.html:183-202; its all-reduces: models/tensor_op.py:
179,326,359.)

Inputs
  --shards  JSONL of tools/tp_shard_bench.py lines (rank 0's shard of a W-way engine on one GPU, whole-forward hipGraphs,
            --local-exchange --gemm-exchange: every exchange runs in its shipped form — inside the o_proj / down_proj
            GEMM, tf_skinny_gemm_xchg — against a one-rank group, so its ON-DEVICE cost (staging store, epoch / flag
            round trips through fine-grained memory) is inside the stage latencies; W = 1 lines are the engine at
            world size 1).
  loop statistics per configuration (tokens per step, inner iterations, host overhead per step) from tracked bench lines
            of the single-GPU run at the same acceptance dial (LOOP below cites the files).

Model (one outer step):
  step(W) = target_verify(W) + k * retrieval_verify(W) + (k + 1) * draft_step + host(W) + X(W) * extra_per_exchange
  X(W)    = exchanges per outer step = 2 L (1 + k)           (two per layer and forward; none at W = 1)
  host(W) = step overhead measured at W = 1 + (k + 1) * 2 small broadcasts (drafted token, decision record) at W > 1
  extra_per_exchange: what a multi-GPU node adds per exchange to the measured one-GPU figure — one scenario per
  exchange form, low / high:
      gemm_exchange         (shipped) per-panel flag hop + one remote-read round trip over xGMI      +2.5 .. +6 us
      exchange_kernel_done  GEMM, then tf_allreduce_oneshot_add_ss as its own launch: +1 .. +3.6 us on the device
                            (profiles/r04_tp_shard_gemm_exchange_ab.jsonl: 0.7 us at 7 rows, 3.6 at 17) + READY hop +
                            remote read + DONE hop                                                     +5.5 .. +14.6 us
      rccl                  GEMM, then a ring all-reduce of 15-25 us instead of the ~4.4 us the fused exchange costs
                            on the device                                                              +10 .. +20 us
  (hop prices: MI355X_MICROARCH.md handoff-flag rows, 2-5 us per cross-device flag hop; the payload — a 16-column panel
   of <= 32 rows from each of W - 1 peers per workgroup — is one round trip of 8-byte loads issued together, not bandwidth.)

    python tools/predict_scaling.py --shards profiles/r04_tp_shard_by_world.jsonl --out profiles/r04_predicted_scaling.json
"""
import argparse
import json

# loop statistics at the aligned 0.7 / 0.9 acceptance dial, single-GPU runs of this repo (tracked files)
LOOP = {
    "configs[1]": {"match": {"target": "llama-7B-128K", "prefill": 124928, "budget": 4096, "gamma": 6},
                   "tokens_per_step": 5.85, "inner_iterations": 3.9, "host_overhead_us": 400.0,
                   "source": "profiles/r03_bench_default_n1.json (20 steps; BENCH_r03.json: 5.85 tokens per step, 411 us step overhead)"},
    "configs[3]": {"match": {"target": "llama-7B-128K", "prefill": 130048, "budget": 12288, "gamma": 16},
                   "tokens_per_step": 10.25, "inner_iterations": 9.5, "host_overhead_us": 1100.0,
                   "source": "profiles/r03_bench_offload_cfg3_world1.json (10.25 tokens per step); inner iterations and "
                             "overhead taken from the gamma = 16 line r03_bench_13b_cfg4_world1.json; all layers HBM-resident "
                             "(the on_chip = 9 offloading tier is PCIe-bound by construction: DESIGN section 6)"},
    "configs[4]": {"match": {"target": "llama-13B-128K", "prefill": 130048, "budget": 12288, "gamma": 16},
                   "tokens_per_step": 7.1, "inner_iterations": 9.5, "host_overhead_us": 1137.0,
                   "source": "profiles/r03_bench_13b_cfg4_world1.json (7.1 tokens per step, 9.5 inner iterations, 1 137 us overhead)"},
}
SCENARIOS = {"gemm_exchange": (2.5, 6.0), "exchange_kernel_done": (5.5, 14.6), "rccl": (10.0, 20.0)}
BCAST_US = (8.0, 20.0)            # one small RCCL broadcast, low / high; two per inner iteration and outer step at W > 1


def predict(line, loop, extra, bcast):
    W, L = line["emulated_world"], line["layers"]
    k = loop["inner_iterations"]
    tv, rv, dr = line["target_verify_us"], line["retrieval_verify_us"], line["draft_step_us"]
    x = 0 if W == 1 else 2 * L * (1 + k)
    host = loop["host_overhead_us"] + (0 if W == 1 else 2 * (k + 1) * bcast)
    terms = {"target_verify": tv, "retrieval_verify": k * rv, "draft": (k + 1) * dr, "host": host, "exchange_xgmi": x * extra}
    step = sum(terms.values())
    return step, terms, x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", required=True)
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    lines = [json.loads(l) for l in open(args.shards) if l.startswith("{") and "emulated_world" in l]
    out = {"model": __doc__.split("Model (one outer step")[1].split("python tools")[0].strip(), "scenarios_us_per_exchange": SCENARIOS,
           "broadcast_us": BCAST_US, "configs": {}}
    md = []
    for name, loop in LOOP.items():
        rows = sorted((l for l in lines if all(l.get(k) == v for k, v in loop["match"].items())),
                      key=lambda l: l["emulated_world"])
        if not rows:
            continue
        cfg = {"loop": {k: v for k, v in loop.items() if k != "match"}, "measured_per_rank_us": [], "predictions": {}}
        for l in rows:
            cfg["measured_per_rank_us"].append({k: l[k] for k in ("emulated_world", "heads_per_rank", "draft_step_us",
                                                                   "retrieval_verify_us", "target_verify_us", "decode_layer",
                                                                   "exchange") if k in l})
        base = None
        for scen, (lo, hi) in SCENARIOS.items():
            preds = []
            for l in rows:
                p = {}
                for tag, extra, bc in (("low", lo, BCAST_US[0]), ("high", hi, BCAST_US[1])):
                    step, terms, x = predict(l, loop, extra, bc)
                    p[tag] = {"ms_per_step": round(step / 1e3, 3), "tokens_per_s": round(loop["tokens_per_step"] / step * 1e6, 1),
                              "dominant": max(terms, key=terms.get), "terms_ms": {k: round(v / 1e3, 3) for k, v in terms.items()},
                              "exchanges_per_step": round(x)}
                if l["emulated_world"] == 1:
                    base = p["low"]["tokens_per_s"]
                for tag in ("low", "high"):
                    if base:
                        p[tag]["speedup_vs_w1"] = round(p[tag]["tokens_per_s"] / base, 2)
                        p[tag]["efficiency"] = round(p[tag]["tokens_per_s"] / base / l["emulated_world"], 3)
                preds.append({"world": l["emulated_world"], **p})
            cfg["predictions"][scen] = preds
        out["configs"][name] = cfg
        md.append(f"**{name}** ({loop['tokens_per_step']} tokens per step, {loop['inner_iterations']} inner iterations)")
        md.append("| W | per rank: target / retrieval verify / draft (us) | GEMM + exchange (shipped): tokens/s (x W=1, eff.) | exchange kernel + DONE | RCCL | dominant term |")
        md.append("|---|---|---|---|---|---|")
        for i, l in enumerate(rows):
            cells = []
            for scen in SCENARIOS:
                p = cfg["predictions"][scen][i]
                if l["emulated_world"] == 1:
                    cells.append(f"{p['low']['tokens_per_s']}")
                else:
                    cells.append(f"{p['high']['tokens_per_s']}-{p['low']['tokens_per_s']} "
                                 f"({p['high'].get('speedup_vs_w1', '-')}-{p['low'].get('speedup_vs_w1', '-')}x, "
                                 f"{p['high'].get('efficiency', '-')}-{p['low'].get('efficiency', '-')})")
            dom = cfg["predictions"]["gemm_exchange"][i]["high"]["dominant"]
            md.append(f"| {l['emulated_world']} | {l['target_verify_us']:.0f} / {l['retrieval_verify_us']:.0f} / {l['draft_step_us']:.0f} | "
                      + " | ".join(cells) + f" | {dom} |")
        md.append("")
    out["markdown"] = "\n".join(md)
    json.dump(out, open(args.out, "w"), indent=1)
    print(out["markdown"])


if __name__ == "__main__":
    main()
