"""Per-(kernel, launch grid, DURATION CLUSTER) statistics from a rocprofv3 --kernel-trace CSV.

The split-KV attention kernel serves the target verify / autoregressive step (125K keys, ~340 us) and the retrieval
verify (4 103 keys, ~20 us) with the SAME name and — since the one-workgroup-per-CU split rule — the SAME launch grid, so
neither `--stats` nor a by-grid split separates them.  Their durations differ 15x: every (kernel, grid) group is cut
into clusters wherever two consecutive sorted durations differ by more than 2.5x (clusters of < 8 launches are merged
into their neighbour), and the attention clusters are priced against the HBM roofline with the algorithmic bytes of
SURVEY.md section 8(d): 2 * keys * heads * head_dim * 2 bytes per launch.

Usage: python tools/attn_by_grid.py <kernel_trace.csv> <out.json> "<source note>"
           [--heads 32 --dim 128 --keys-target 124990 --keys-retrieval 4103]
"""
import argparse
import csv
import json
from collections import defaultdict

HBM_PEAK_GBPS = 8000.0


def clusters(v, ratio=2.5, min_size=8):
    v = sorted(v)
    cuts = [0] + [i for i in range(1, len(v)) if v[i] > ratio * max(v[i - 1], 0.5)] + [len(v)]
    parts = [v[a:b] for a, b in zip(cuts, cuts[1:]) if b > a]
    merged = []
    for p in parts:                                 # stragglers (a handful of outliers) join the previous cluster
        if merged and len(p) < min_size:
            merged[-1] = merged[-1] + p
        elif merged and len(merged[-1]) < min_size:
            merged[-1] = merged[-1] + p
        else:
            merged.append(p)
    return merged


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("out")
    ap.add_argument("note", nargs="?", default="")
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--keys-target", type=int, default=124990, help="mean key count of a target-verify launch")
    ap.add_argument("--keys-retrieval", type=int, default=4103)
    a = ap.parse_args()
    rows = defaultdict(list)
    with open(a.trace) as f:
        for r in csv.DictReader(f):
            n = r["Kernel_Name"]
            if not any(s in n for s in ("attn_", "skinny_gemm", "topp", "retrieval_", "accept", "sample_")):
                continue
            key = (n[:64], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Grid_Size_Y", 0) or 0))
            rows[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = []
    for (n, gx, gy), v in rows.items():
        for c in clusters(v):
            row = {"kernel": n, "grid_x": gx, "grid_y": gy, "calls": len(c), "total_ms": round(sum(c) / 1e3, 3),
                   "avg_us": round(sum(c) / len(c), 2), "median_us": round(c[len(c) // 2], 2), "min_us": round(c[0], 2),
                   "max_us": round(c[-1], 2)}
            if "attn_split" in n:                      # price the attention clusters against the HBM roofline
                med = c[len(c) // 2]
                if med > 100.0:
                    stage, keys = "target verify / autoregressive step (full KV)", a.keys_target
                elif med > 12.0:
                    stage, keys = "retrieval verify (retrieval cache)", a.keys_retrieval
                else:
                    stage, keys = "short stream (probe / draft-sized)", None
                row["stage"] = stage
                if keys:
                    byts = 2 * keys * a.heads * a.dim * 2
                    gbps = byts / (row["avg_us"] * 1e-6) / 1e9
                    row.update(algorithmic_bytes_per_launch=byts, achieved_GBps=round(gbps, 1),
                               frac_of_hbm_peak=round(gbps / HBM_PEAK_GBPS, 4))
            out.append(row)
    out.sort(key=lambda r: -r["total_ms"])
    json.dump({"source": a.note, "clustering": "per (kernel, grid): cut where consecutive sorted durations differ > 2.5x",
               "hbm_peak_GBps": HBM_PEAK_GBPS, "rows": out[:48]}, open(a.out, "w"), indent=1)
    print(json.dumps([r for r in out if "stage" in r] + out[:6], indent=1))


if __name__ == "__main__":
    main()
