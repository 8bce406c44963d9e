"""The measurement tools that turn a rocprofv3 kernel trace into the tracked evidence (profiles/*by_stage*.json,
*gap_analysis*.txt) on synthetic traces: the duration clustering must separate the split-KV kernel's target-verify and
retrieval-verify launches (same name, same grid) and price them with the SURVEY 8(d) bytes; the gap analysis must window
exactly N outer steps."""
import csv
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = ["Kernel_Name", "Grid_Size_X", "Grid_Size_Y", "Start_Timestamp", "End_Timestamp"]


def _write(path, rows):
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=FIELDS)
        w.writeheader()
        for r in rows:
            w.writerow(dict(zip(FIELDS, r)))


def test_attn_by_grid_separates_and_prices_the_two_verify_stages(tmp_path):
    name = "_Z17attn_split_kernelILi128ELi1EEvPKDF16_S1_S1_lliiPKiifiPfPjPDF16_"
    rows, t = [], 0
    for i in range(64):                                   # 64 target-verify launches ~340 us, 256 retrieval ~20 us
        rows.append((name, 2048, 32, t, t + 340_000 + 1000 * (i % 7)))
        t += 400_000
        for j in range(4):
            rows.append((name, 2048, 32, t, t + 20_000 + 200 * j))
            t += 30_000
    rows.append((name, 2048, 32, t, t + 7_000))            # a lone short probe launch: merged into its neighbour cluster
    rows.append(("_Z18skinny_gemm_kernelILi1ELi1ELb1ELi4EE", 176128, 1, t + 10_000, t + 42_000))
    trace, out = tmp_path / "trace.csv", tmp_path / "out.json"
    _write(trace, rows)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "attn_by_grid.py"), str(trace), str(out), "synthetic",
                    "--keys-target", "124936", "--keys-retrieval", "4103"], check=True, capture_output=True)
    res = json.load(open(out))
    stages = {r["stage"]: r for r in res["rows"] if "stage" in r}
    tv = stages["target verify / autoregressive step (full KV)"]
    rv = stages["retrieval verify (retrieval cache)"]
    assert tv["calls"] == 64 and rv["calls"] == 257
    assert tv["algorithmic_bytes_per_launch"] == 2 * 124936 * 32 * 128 * 2
    assert abs(tv["achieved_GBps"] - tv["algorithmic_bytes_per_launch"] / (tv["avg_us"] * 1e-6) / 1e9) < 0.1
    assert 0.70 < tv["frac_of_hbm_peak"] < 0.78 and 0.38 < rv["frac_of_hbm_peak"] < 0.45


def test_gap_analysis_windows_exactly_n_outer_steps(tmp_path):
    rows, t = [], 0
    for step in range(6):                                  # each outer step: busy 10 ms, one 50 us idle gap, then the accept
        rows.append(("verify_kernel", 1, 1, t, t + 10_000_000))
        t += 10_000_000 + 50_000
        rows.append(("accept_chain_kernel(float const*)", 1, 1, t, t + 10_000))
        t += 10_000
    rows.append(("probe_kernel", 1, 1, t + 5_000_000, t + 6_000_000))   # trailing probes without accept_chain launches
    trace = tmp_path / "trace.csv"
    _write(trace, rows)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gap_analysis.py"), str(trace), "--steps", "4"],
                         check=True, capture_output=True, text=True).stdout
    assert "window = 4 outer steps" in out
    first = [ln for ln in out.splitlines() if ln.startswith("window ") and "ms:" in ln][0]
    span_ms = float(first.split()[1])
    assert abs(span_ms - 4 * 10.06) < 0.05, first          # 4 x (10 ms + 50 us + 10 us), the trailing probe excluded
    assert "idle 0.20 ms" in first, first


def test_pmc_layer_reduce_matches_dispatches_to_the_launch_list(tmp_path):
    """tools/pmc_layer_reduce.py on a synthetic pair of counter files: per-XCC rows are summed per dispatch, kernels that
    are not part of the layer are ignored, the first launch per label is dropped, FETCH_SIZE is doubled, and a dispatch
    order that disagrees with the launch list is an error rather than a wrong table."""
    import csv
    from tools import pmc_layer_reduce as R
    launches = [{"label": "a", "kernel": "skinny_gemm", "algorithmic_bytes": 2048 * 1024}] * 3 \
        + [{"label": "b", "kernel": "attn_split", "algorithmic_bytes": 1024 * 1024}] * 2
    meta = {"rows": 7, "hidden": 4096, "inter": 11008, "heads": 32, "head_dim": 128, "slots": 4103, "launches": launches}

    def write(sub, counter, val, names):
        d = tmp_path / sub / "x"
        d.mkdir(parents=True)
        with open(d / "1_counter_collection.csv", "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
            for i, name in enumerate(names):
                w.writerow([i + 1, name, counter, val / 2])
                w.writerow([i + 1, name, counter, val / 2])
            w.writerow([99, "some_other_kernel", counter, 5])
        return str(tmp_path / sub)
    names = ["skinny_gemm_kernel<1>"] * 3 + ["attn_split_kernel<128>"] * 2
    out = R.reduce(write("f", "FETCH_SIZE", 1024.0, names), write("w", "WRITE_SIZE", 8.0, names), meta)
    a, b = out["kernels"]
    assert a["launches_counted"] == 2 and a["hbm_read_bytes_per_launch_corrected"] == 2 * 1024 * 1024
    assert a["traffic_over_algorithmic"] == round((2 * 1024 * 1024 + 8192) / (2048 * 1024), 4)
    assert b["launches_counted"] == 1 and b["traffic_over_algorithmic"] == round((2 * 1024 * 1024 + 8192) / (1024 * 1024), 4)
    bad = write("f2", "FETCH_SIZE", 1024.0, names[::-1])
    with pytest.raises(SystemExit):
        R.reduce(bad, str(tmp_path / "w"), meta)


def test_oneshot_allreduce_alternating_halves_host_bookkeeping():
    """The host side of tf_allreduce_oneshot_alt without a device: exchange e (1-based) must be staged in half e & 1,
    `staging()` must hand out that half, `is_staged()` must accept only it, and a reduce advances the count — the
    invariant the kernel's epoch / half guard (sticky error 3) checks on the device."""
    import ctypes
    import torch
    from triforce_amd.utils import oneshot_ar as M

    calls = []

    class Lib:
        def tf_allreduce_oneshot_alt(self, *a):
            calls.append(a)
            return 0

    ar = object.__new__(M.OneShotAllReduce)
    ar.rank, ar.world, ar.device, ar.max_elems = 0, 1, torch.device("cpu"), 64
    ar.alternate, ar._issued = True, 0
    ar._stage = torch.zeros(128, dtype=torch.float16)
    ar.data_ptr, ar.flags_ptr = ar._stage.data_ptr(), 0
    ar._data = (ctypes.c_void_p * 1)(ar.data_ptr)
    ar._flags = (ctypes.c_void_p * 1)(0)
    old_lib, old_stream = M.hip.lib, torch.cuda.current_stream
    M.hip.lib = lambda: Lib()
    torch.cuda.current_stream = lambda device=None: type("S", (), {"cuda_stream": 0})()
    try:
        halves = []
        for e in range(1, 6):
            st = ar.staging(2, 8)
            assert ar.is_staged(st) and st.data_ptr() == ar.data_ptr + 2 * 64 * (e & 1)
            other = ar._stage[64 * ((e + 1) & 1):][:16].view(2, 8)
            assert not ar.is_staged(other)
            out = torch.zeros(2, 8, dtype=torch.float16)
            ar.reduce(st, out)
            halves.append(calls[-1][-2])                       # expect_half as passed to the C entry
            assert calls[-1][-3] == 64 and ar._issued == e     # half_elems, exchanges counted
        assert halves == [1, 0, 1, 0, 1]
    finally:
        M.hip.lib, torch.cuda.current_stream = old_lib, old_stream


def test_predict_scaling_composes_stage_latencies_and_exchange_prices(tmp_path):
    """tools/predict_scaling.py: step(W) = target verify + k * retrieval verify + (k + 1) * draft + host + X * extra with
    X = 2 L (1 + k) exchanges at W > 1 — checked by hand on two synthetic shard lines; W = 1 pays no exchange, the RCCL
    scenario is slower than the one-shot ones, efficiency = speed-up / W."""
    lines = [{"target": "llama-7B-128K", "emulated_world": 1, "heads_per_rank": 32, "layers": 32, "prefill": 124928,
              "budget": 4096, "gamma": 6, "draft_step_us": 120.0, "retrieval_verify_us": 3300.0, "target_verify_us": 13700.0},
             {"target": "llama-7B-128K", "emulated_world": 8, "heads_per_rank": 4, "layers": 32, "prefill": 124928,
              "budget": 4096, "gamma": 6, "draft_step_us": 130.0, "retrieval_verify_us": 1900.0, "target_verify_us": 3300.0}]
    shards, out = tmp_path / "shards.jsonl", tmp_path / "pred.json"
    shards.write_text("\n".join(json.dumps(l) for l in lines))
    # (no bench line of the round for this configuration: the embedded fall-back statistics and the shard file's own W = 1 row;
    #  with one, the W = 1 row is the single-GPU graph engine — tests/test_host_edges_cpu.py covers that)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "predict_scaling.py"), "--shards", str(shards), "--out",
                    str(out), "--bench", f"configs[1]={tmp_path / 'no_such_bench.json'}", "--tp-host",
                    f"configs[1]={tmp_path / 'no_such_tp_bench.json'}"], check=True, capture_output=True)
    res = json.load(open(out))
    cfg = res["configs"]["configs[1]"]
    k, tps, host = cfg["loop"]["inner_iterations"], cfg["loop"]["tokens_per_step"], cfg["loop"]["host_overhead_us"]
    w1 = cfg["predictions"]["gemm_exchange"][0]["low"]
    step1 = 13700.0 + k * 3300.0 + (k + 1) * 120.0 + host
    assert abs(w1["ms_per_step"] - step1 / 1e3) < 1e-3 and w1["exchanges_per_step"] == 0
    assert abs(w1["tokens_per_s"] - tps / step1 * 1e6) < 0.1
    w8 = cfg["predictions"]["gemm_exchange"][1]
    x = 2 * 32 * (1 + k)
    lo, hi = res["scenarios_us_per_exchange"]["gemm_exchange"]
    tp_extra = cfg["loop"]["tp_host_extra_us"]                 # what the TP engine's loop adds on the host at W > 1
    assert tp_extra >= 0
    step8 = 3300.0 + k * 1900.0 + (k + 1) * 130.0 + host + tp_extra + 2 * (k + 1) * res["broadcast_us"][0] + x * lo
    assert abs(w8["low"]["ms_per_step"] - step8 / 1e3) < 1e-3 and w8["low"]["exchanges_per_step"] == round(x)
    assert w8["high"]["tokens_per_s"] < w8["low"]["tokens_per_s"]
    assert abs(w8["low"]["efficiency"] - w8["low"]["speedup_vs_w1"] / 8) < 2e-3
    rccl8 = cfg["predictions"]["rccl"][1]
    assert rccl8["low"]["tokens_per_s"] < w8["low"]["tokens_per_s"]
    assert "| 8 |" in res["markdown"] and "configs[1]" in res["markdown"]


def test_kernel_timeline_groups_by_kernel_and_grid_and_measures_gaps(tmp_path):
    rows, t = [], 0
    for i in range(20):
        rows.append(("gemm_a", 256, 1, t, t + 5_000))
        t += 5_000 + 1_500                                   # 1.5 us boundary
        rows.append(("gemm_a", 96, 1, t, t + 9_000))         # same name, other grid: its own row
        t += 9_000 + 100_000                                  # a 100 us host gap: counted as a long gap, not averaged
    trace, out = tmp_path / "trace.csv", tmp_path / "tl.json"
    _write(trace, rows)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_timeline.py"), str(trace), str(out), "synthetic"],
                   check=True, capture_output=True)
    ks = {r["kernel"]: r for r in json.load(open(out))["kernels"]}
    a, b = ks["gemm_a | grid 256 wg 0"], ks["gemm_a | grid 96 wg 0"]
    assert a["launches"] == 20 and abs(a["avg_us"] - 5.0) < 1e-6 and a["long_gaps"] == 19
    assert b["launches"] == 20 and abs(b["avg_us"] - 9.0) < 1e-6 and abs(b["avg_gap_before_us"] - 1.5) < 1e-6


def test_build_variant_command_line(monkeypatch):
    """triforce_amd.build.build_variant: "NAME=VALUE" -> -DNAME=VALUE, "-flag" passed through, "!kernarg-preload" builds
    without the -mllvm -amdgpu-kernarg-preload-count pair; the default build carries the pair (csrc/gemv.hip orders its
    kernel arguments for it)."""
    from triforce_amd import build as tb
    seen = []
    monkeypatch.setattr(tb.subprocess, "run", lambda cmd, check=True: seen.append(list(cmd)))
    monkeypatch.setattr(tb, "hipcc_path", lambda: "hipcc")
    assert "-amdgpu-kernarg-preload-count=14" in tb.FLAGS and tb.FLAGS[tb.FLAGS.index("-amdgpu-kernarg-preload-count=14") - 1] == "-mllvm"
    out = tb.build_variant("t1", ["SG_LN_PRE=1", "-save-temps"], verbose=False)
    assert out.endswith("libtriforce_hip_t1.so")
    cmd = seen[-1]
    assert "-DSG_LN_PRE=1" in cmd and "-save-temps" in cmd and "-amdgpu-kernarg-preload-count=14" in cmd
    assert cmd[-2:] == ["-o", out] and any(c.endswith("gemv.hip") for c in cmd)
    tb.build_variant("t2", ["!kernarg-preload"], verbose=False)
    cmd = seen[-1]
    assert "-mllvm" not in cmd and not any(c.startswith("-amdgpu-kernarg-preload") for c in cmd)
    assert "--offload-arch=gfx950" in cmd and not any(c.startswith("-D!") or c.startswith("!") for c in cmd)


def test_parity_table_folds_recorded_checks_into_the_tolerance_table(tmp_path):
    """tools/parity_table.py: the worst use of a bound over a test's parametrisations, slack = 1 / used, and the
    list of checks with >= 4x slack."""
    import json
    import subprocess
    import sys
    rows = [{"test": "tests/test_x.py::test_a[1]", "check": "assert_close", "used": 0.2, "what": "", "max_abs": 2e-4, "mean_abs": 1e-6, "atol": 1e-3, "rtol": 1e-3, "max_ref": 1.0},
            {"test": "tests/test_x.py::test_a[2]", "check": "assert_close", "used": 0.6, "what": "", "max_abs": 6e-4, "mean_abs": 2e-6, "atol": 1e-3, "rtol": 1e-3, "max_ref": 1.0},
            {"test": "tests/test_x.py::test_b", "check": "bound", "used": 0.1, "what": "rows", "value": 1e-3, "limit": 1e-2},
            {"test": "tests/test_x.py::test_c", "check": "logit_check", "used": 0.75, "what": "logits", "max_abs": 5.9e-3, "mean_abs": 6e-4, "bound": 7.8e-3,
             "mean_bound": 1e-3, "mean_used": 0.6, "max_ref": 7.0, "spacings": 2.0}]
    src, dst, js = tmp_path / "t.jsonl", tmp_path / "t.md", tmp_path / "t.json"
    src.write_text("\n".join(json.dumps(r) for r in rows) + "\n")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "parity_table.py"), str(src), str(dst), "--json", str(js)], check=True)
    table = {t["test"]: t for t in json.load(open(js))}
    assert table["test_x.py::test_a"]["cases"] == 2 and abs(table["test_x.py::test_a"]["used"] - 0.6) < 1e-9
    assert table["test_x.py::test_b — rows"]["slack"] == 10.0 and table["test_x.py::test_c — logits"]["slack"] == 1.33
    assert "1 with >= 4x slack: test_x.py::test_b — rows (10.0x)" in dst.read_text()


@pytest.mark.parametrize("probe", sorted(p for p in os.listdir(os.path.join(ROOT, "tools", "probes")) if p.endswith(".hip")))
def test_stand_alone_probes_still_compile_for_gfx950(probe, tmp_path):
    """tools/probes/*.hip are the stand-alone measurements DESIGN cites (seams, hardware transpose reads): they must keep building."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", os.path.join(ROOT, "tools", "probes", probe),
                          "-o", str(tmp_path / "probe.o")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
