"""Idle-gap analysis of a rocprofv3 --kernel-trace CSV: GPU busy time, idle time and the largest sources of idle (which
kernel follows the gap) over a window of the run.
Usage: python tools/gap_analysis.py <kernel_trace.csv> [tail_ms]            the last tail_ms milliseconds of the run
       python tools/gap_analysis.py <kernel_trace.csv> --steps N [--skip M]  exactly N outer TriForce steps: the window runs
           from the end of one accept_chain_kernel (the last kernel decision of an outer step) to the end of the N-th one
           after it, counted back from the LAST accept_chain_kernel of the run minus M (bench.py's timed steps are followed
           by probes without accept_chain launches, so M = 0 selects the last N timed steps)"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
if "--steps" in sys.argv:
    n = int(sys.argv[sys.argv.index("--steps") + 1])
    skip = int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 0
    marks = [e for s, e, name in rows if "accept_chain_kernel" in name]
    if len(marks) < n + skip + 1:
        raise SystemExit(f"only {len(marks)} accept_chain_kernel launches in the trace")
    hi = marks[len(marks) - 1 - skip]
    lo = marks[len(marks) - 1 - skip - n]
    rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
    print(f"window = {n} outer steps (between accept_chain_kernel launches)")
else:
    tail_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
    t_end = rows[-1][1]
    rows = [r for r in rows if r[0] >= t_end - tail_ms * 1e6]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = defaultdict(lambda: [0, 0])
prev_end = rows[0][1]
prev_name = rows[0][2]
big = []
for s, e, n in rows[1:]:
    g = s - prev_end
    if g > 0:
        key = (prev_name[:48], n[:48])
        gaps[key][0] += g
        gaps[key][1] += 1
        if g > 20000:
            big.append((g, prev_name[:40], n[:40]))
    prev_end = max(prev_end, e)
    prev_name = n
print(f"window {span/1e6:.2f} ms: busy {busy/1e6:.2f} ms ({100*busy/span:.1f}%), idle {(span-busy)/1e6:.2f} ms, {len(rows)} kernels")
print("top idle sources (after-kernel -> before-kernel): total_us count avg_us")
for (a, b), (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {t/1e3:9.1f} {c:6d} {t/c/1e3:7.2f}   {a}  ->  {b}")
print("gaps > 20 us:", len(big), "total", sum(g for g, _, _ in big) / 1e3, "us")
bytime = defaultdict(lambda: [0, 0])
for s, e, n in rows:
    bytime[n[:70]][0] += e - s
    bytime[n[:70]][1] += 1
print("top kernels in window:")
for n, (t, c) in sorted(bytime.items(), key=lambda kv: -kv[1][0])[:18]:
    print(f"  {t/1e3:9.1f} us {c:6d} x {t/c/1e3:8.2f}   {n}")
