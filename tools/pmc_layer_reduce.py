"""Reduce the two rocprofv3 --pmc passes over tools/pmc_layer.py to HBM traffic per launch for every kernel of the
retrieval-verify decoder layer.
    python tools/pmc_layer_reduce.py <FETCH_SIZE dir> <WRITE_SIZE dir> <meta.json> <out.json> [source note]
The dispatches of the skinny GEMM / split-KV attention kernels are matched, in dispatch order, with the launch list the
driver wrote.  Units as in tools/pmc_reduce.py: KiB, FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM section).
The first launch of each label is dropped (cold instruction cache / first touch of the inputs)."""
import csv
import glob
import json
import os
import sys


def dispatches(d, counter, substrs):
    by = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name") or row.get("Kernel Name") or ""
                if (row.get("Counter_Name") or row.get("Counter Name")) != counter:
                    continue
                hit = [s for s in substrs if s in name]
                if not hit:
                    continue
                did = int(row.get("Dispatch_Id") or row.get("Dispatch Id") or 0)
                e = by.setdefault(did, [hit[0], 0.0])
                e[1] += float(row.get("Counter_Value") or row.get("Counter Value"))    # per-XCC rows: sum per dispatch
    return [by[k] for k in sorted(by)]


def reduce(fdir, wdir, meta, note=""):
    launches = meta["launches"]
    subs = sorted({l["kernel"] for l in launches})
    fetch, write = dispatches(fdir, "FETCH_SIZE", subs), dispatches(wdir, "WRITE_SIZE", subs)
    if len(fetch) != len(launches) or len(write) != len(launches):
        raise SystemExit(f"{len(launches)} launches listed, {len(fetch)} / {len(write)} dispatches counted")
    rows = {}
    for l, (kf, f), (kw, w) in zip(launches, fetch, write):
        if kf != l["kernel"] or kw != l["kernel"]:
            raise SystemExit(f"dispatch order does not match the launch list at {l['label']}: {kf} / {kw}")
        rows.setdefault(l["label"], {"algorithmic_bytes": l["algorithmic_bytes"], "fetch_KiB": [], "write_KiB": []})
        rows[l["label"]]["fetch_KiB"].append(f)
        rows[l["label"]]["write_KiB"].append(w)
    out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python tools/pmc_layer.py"
                     + (f"  [{note}]" if note else ""),
           "shape": {k: meta[k] for k in ("rows", "hidden", "inter", "heads", "head_dim", "slots")},
           "notes": "KiB counters; FETCH_SIZE doubled (gfx950, MI355X_MICROARCH.md HBM section); first launch per kernel "
                    "dropped; every launch reads a different weight / KV copy (> 600 MB per shape in rotation)",
           "kernels": []}
    for label, r in rows.items():
        f, w = r["fetch_KiB"][1:], r["write_KiB"][1:]
        rd, wr = sum(f) / len(f) * 1024 * 2, sum(w) / len(w) * 1024
        out["kernels"].append({"kernel": label, "algorithmic_bytes_per_launch": r["algorithmic_bytes"],
                               "hbm_read_bytes_per_launch_corrected": int(rd), "hbm_write_bytes_per_launch": int(wr),
                               "traffic_over_algorithmic": round((rd + wr) / r["algorithmic_bytes"], 4),
                               "launches_counted": len(f)})
    return out


if __name__ == "__main__":
    fdir, wdir, metap, outp = sys.argv[1:5]
    res = reduce(fdir, wdir, json.load(open(metap)), sys.argv[5] if len(sys.argv) > 5 else "")
    with open(outp, "w") as fh:
        json.dump(res, fh, indent=1)
    for k in res["kernels"]:
        print(k["kernel"], k["traffic_over_algorithmic"])
