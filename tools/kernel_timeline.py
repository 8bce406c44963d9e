"""Per-kernel durations AND the idle gap in front of each kernel from a rocprofv3 --kernel-trace CSV, grouped by (kernel,
grid, workgroup): what a decode layer's time is made of — kernel bodies vs dependent-launch boundaries.

    python tools/kernel_timeline.py <kernel_trace.csv> <out.json> "<note>" [--min-count N]

Gaps are measured to the END of the previous kernel in start order; gaps above 50 us (host round trips, graph-launch
boundaries) are counted separately and not averaged in."""
import csv
import json
import re
import sys
from collections import defaultdict

path, out_path, note = sys.argv[1], sys.argv[2], sys.argv[3]
min_count = int(sys.argv[sys.argv.index("--min-count") + 1]) if "--min-count" in sys.argv else 8
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], grid, wg))
rows.sort()


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:90]


agg = defaultdict(lambda: {"n": 0, "dur": 0, "gap": 0, "gap_n": 0, "long_gaps": 0, "prev": defaultdict(int)})
prev_end, prev_key = None, None
for s, e, name, grid, wg in rows:
    key = f"{short(name)} | grid {grid} wg {wg}"
    a = agg[key]
    a["n"] += 1
    a["dur"] += e - s
    if prev_end is not None:
        g = s - prev_end
        if g > 50000:
            a["long_gaps"] += 1
        else:
            a["gap"] += max(g, 0)
            a["gap_n"] += 1
            a["prev"][prev_key] += 1
    prev_end, prev_key = max(e, prev_end or 0), key
res = []
for key, a in agg.items():
    if a["n"] < min_count:
        continue
    prev = max(a["prev"].items(), key=lambda kv: kv[1])[0] if a["prev"] else None
    res.append({"kernel": key, "launches": a["n"], "avg_us": round(a["dur"] / a["n"] / 1e3, 2),
                "avg_gap_before_us": round(a["gap"] / max(a["gap_n"], 1) / 1e3, 2), "long_gaps": a["long_gaps"],
                "total_ms": round(a["dur"] / 1e6, 3), "usually_after": prev})
res.sort(key=lambda r: -r["total_ms"])
json.dump({"source": note, "kernels": res}, open(out_path, "w"), indent=1)
for r in res[:40]:
    print(f'{r["avg_us"]:9.2f} us  gap {r["avg_gap_before_us"]:6.2f}  x{r["launches"]:<6} {r["kernel"]}')
