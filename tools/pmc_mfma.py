"""Post-process a rocprofv3 --pmc counter_collection.csv: per-kernel sums of the collected counters."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print(k, {c: (v / max(cnt[(k, c)], 1)) for c, v in d.items()})
