"""Per-(kernel, launch grid) durations from a rocprofv3 --kernel-trace CSV: the split-KV attention kernel serves the
target verify (grid 32 splits x 32 heads) and the retrieval verify (8 x 32) — the --stats average mixes them.
Usage: python tools/attn_by_grid.py <kernel_trace.csv> <out.json> "<source note>" """
import csv
import json
import sys
from collections import defaultdict

rows = defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        if "attn_" not in n and "skinny_gemm" not in n and "topp" not in n:
            continue
        key = (n[:64], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Grid_Size_Y", 0) or 0))
        rows[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = []
for (n, gx, gy), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    out.append({"kernel": n, "grid_x": gx, "grid_y": gy, "calls": len(v), "avg_us": round(sum(v) / len(v), 2),
                "median_us": round(v[len(v) // 2], 2), "min_us": round(v[0], 2), "max_us": round(v[-1], 2)})
json.dump({"source": sys.argv[3] if len(sys.argv) > 3 else "", "rows": out[:40]}, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out[:8], indent=1))
