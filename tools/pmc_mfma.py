"""MFMA-pipe utilisation of the target-verify attention kernel from a rocprofv3 --pmc pass over tools/pmc_attn.py
(north_star: "MFMA utilisation on verify against chip peak").

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES \
              --output-format csv -d <dir> -- python tools/pmc_attn.py          (counters only: no --stats / --sys-trace)
    python tools/pmc_mfma.py <dir> <out.json> ["source note"]

utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (256 CUs x 4 SIMDs x active cycles), active cycles = GRBM_GUI_ACTIVE / 8 XCDs.
The split-KV kernel issues 20 MFMAs per 16-key tile and q-tile, 12 of them QK^T / PV and 8 the on-matrix-core transpose
of V, so the 'useful' figure is 12/20 of the busy one."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    d, out = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    per = defaultdict(lambda: defaultdict(float))            # dispatch -> counter -> value (summed over XCC / SE rows)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if "attn_split" not in (r.get("Kernel_Name") or ""):
                    continue
                per[int(r.get("Dispatch_Id") or 0)][r["Counter_Name"]] += float(r["Counter_Value"])
    if not per:
        raise SystemExit(f"no attn_split dispatches with counters under {d}")
    names = sorted({c for v in per.values() for c in v})
    mean = {c: sum(v.get(c, 0.0) for v in per.values()) / len(per) for c in names}
    res = {"source": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -- python "
                     "tools/pmc_attn.py; counters only" + (f"  [{note}]" if note else ""),
           "kernel": "attn_split_kernel<128,1> (target-verify attention, BASELINE configs[1] layer shape: 8 queries x 124936 "
                     "keys x 32 heads x 128)", "launches": len(per), "per_launch_mean": mean}
    if mean.get("GRBM_GUI_ACTIVE") and mean.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        active = mean["GRBM_GUI_ACTIVE"] / 8.0
        util = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (256 * 4 * active)
        res.update(active_cycles_per_xcd=active, mfma_utilisation=util, mfma_utilisation_useful=util * 12.0 / 20.0,
                   comment="MFMA busy cycles / (256 CUs x 4 SIMDs x active cycles); arithmetic intensity of the verify is "
                           "q = 8 flop per KV byte against a ridge of ~300, so the matrix core idles while the kernel "
                           "streams KV at the HBM rate (HBM-bound by construction, SURVEY 8d)")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: res.get(k) for k in ("launches", "mfma_utilisation", "mfma_utilisation_useful")}))


if __name__ == "__main__":
    main()
