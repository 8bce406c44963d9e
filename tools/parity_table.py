"""Fold gpurun_out/parity_table.jsonl (tests/helpers.record: one row per tolerance check of the GPU suite) into the
tolerance table of DESIGN section 5: test, kind of check, measured max / mean |d|, the bound the test asserts, how much
of the bound the measurement used and the slack factor (bound / measured).  Rows of one test function (parametrisations,
repeated checks) are merged: the WORST use of the bound is what the table shows.

    python tools/parity_table.py gpurun_out/parity_table.jsonl profiles/r06_parity_table.md [--json profiles/r06_parity_table.json]
"""
import json
import re
import sys
from collections import OrderedDict


def bound_text(r):
    c = r["check"]
    if c == "assert_close":
        return f"atol {r['atol']:g} + rtol {r['rtol']:g}·|ref|"
    if c == "logit_check":
        return f"max ≤ max(1e-3, {r['spacings']:g} fp16 spacings at max|logit|) = {r['bound']:.2e}; mean < {r['mean_bound']:g}"
    if c == "ulp_report":
        return f"≤ {r['ulps']} fp16 ulp + {r['atol']:g}; ≤ {100 * r['frac_allowed']:.2g} % of elements differ"
    return f"< {r['limit']:g}"


def main():
    src, dst = sys.argv[1], sys.argv[2]
    js = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    groups = OrderedDict()
    for line in open(src):
        line = line.strip()
        if not line:
            continue
        r = json.loads(line)
        test = re.sub(r"\[.*\]$", "", r["test"])
        key = (test, r["check"], r.get("what", "") if r["check"] in ("bound", "ulp_report") else "")
        g = groups.setdefault(key, {"rows": 0, "worst": None, "max_abs": 0.0, "mean_abs": 0.0, "mean_used": 0.0, "frac_used": 0.0})
        g["rows"] += 1
        if g["worst"] is None or r["used"] > g["worst"]["used"]:
            g["worst"] = r
        g["max_abs"] = max(g["max_abs"], r.get("max_abs", r.get("value", 0.0)))
        g["mean_abs"] = max(g["mean_abs"], r.get("mean_abs", 0.0))
        g["mean_used"] = max(g["mean_used"], r.get("mean_used", 0.0))
        g["frac_used"] = max(g["frac_used"], r.get("frac_used", 0.0))
    out = ["| test | check | cases | measured max \\|d\\| | measured mean \\|d\\| | bound asserted | bound used (worst case) | slack |",
           "|---|---|---|---|---|---|---|---|"]
    table = []
    for (test, check, what), g in groups.items():
        w = g["worst"]
        used = max(w["used"], g["mean_used"], g["frac_used"])
        slack = (1.0 / used) if used > 0 else float("inf")
        name = test.replace("tests/", "") + (f" — {what}" if what else (f" — {w.get('what')}" if w.get("what") else ""))
        out.append(f"| `{name}` | {check} | {g['rows']} | {g['max_abs']:.3e} | {g['mean_abs']:.3e} | {bound_text(w)} | {used:.3f} | "
                   f"{'exact' if slack == float('inf') else f'{slack:.1f}x'} |")
        table.append({"test": name, "check": check, "cases": g["rows"], "max_abs": g["max_abs"], "mean_abs": g["mean_abs"],
                      "bound": bound_text(w), "used": used, "slack": None if slack == float("inf") else round(slack, 2)})
    loose = [t for t in table if t["slack"] is not None and t["slack"] >= 4.0]
    out.append("")
    out.append(f"{len(table)} checks; {len(loose)} with >= 4x slack" + (": " + "; ".join(f"{t['test']} ({t['slack']}x)" for t in loose) if loose else ""))
    with open(dst, "w") as f:
        f.write("\n".join(out) + "\n")
    if js:
        with open(js, "w") as f:
            json.dump(table, f, indent=1)
    print("\n".join(out[-1:]))


if __name__ == "__main__":
    main()
