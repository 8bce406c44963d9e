"""Reduce the two rocprofv3 --pmc passes over tools/pmc_attn.py (FETCH_SIZE and WRITE_SIZE, collected in separate runs)
to the HBM traffic per launch of the target-verify attention kernel.
    python tools/pmc_reduce.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> [source note]
Units (MI355X_MICROARCH.md, HBM section): both counters are KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide
coalesced streaming read and is doubled here."""
import csv
import glob
import json
import os
import sys


def per_launch(d, counter, kernel_substr="attn_split"):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name") or row.get("Kernel Name") or ""
                if kernel_substr in name and (row.get("Counter_Name") or row.get("Counter Name")) == counter:
                    vals.append((int(row.get("Dispatch_Id") or row.get("Dispatch Id") or 0),
                                 float(row.get("Counter_Value") or row.get("Counter Value"))))
    by = {}
    for did, v in vals:                      # a counter may be reported per XCC / dimension: sum per dispatch
        by[did] = by.get(did, 0.0) + v
    return [by[k] for k in sorted(by)]


if __name__ == "__main__":
    fdir, wdir, out = sys.argv[1:4]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    sq, sk, H, D = 8, 124936, 32, 128
    alg = 2 * sk * H * D * 2
    fetch, write = per_launch(fdir, "FETCH_SIZE"), per_launch(wdir, "WRITE_SIZE")
    fm, wm = sum(fetch) / len(fetch), sum(write) / len(write)
    rd, wr = fm * 1024 * 2, wm * 1024
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) -- python tools/pmc_attn.py"
                     + (f"  [{note}]" if note else ""),
           "shape": {"sq": sq, "sk": sk, "H": H, "D": D}, "algorithmic_bytes_per_launch": alg,
           "notes": "FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE doubled (gfx950 reports half the bytes of a wide coalesced "
                    "streaming read, MI355X_MICROARCH.md HBM section); launches alternate between two KV buffers so the "
                    "256 MiB Infinity Cache cannot serve re-reads",
           "attn_split_kernel": {"FETCH_SIZE": {"per_launch_KiB": fetch, "mean_KiB": fm},
                                 "WRITE_SIZE": {"per_launch_KiB": write, "mean_KiB": wm}},
           "hbm_read_bytes_per_launch_corrected": int(rd), "hbm_write_bytes_per_launch": int(wr),
           "traffic_bytes_per_launch": int(rd + wr), "traffic_over_algorithmic": round((rd + wr) / alg, 4)}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: res[k] for k in ("traffic_bytes_per_launch", "traffic_over_algorithmic")}))
