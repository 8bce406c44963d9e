"""What a tensor-parallel rank's norm-prologue GEMM spends its 10-14 us on: phase stamps of EVERY workgroup of
one launch, from the instrumented build of the skinny GEMM (-DSG_STAMPS=1: wave 0 of each workgroup stores s_memtime at
entry / prologue loads issued / norm scale known / first k-batch done / K loop done / waves merged / exit, plus the 100 MHz
wall clock at entry and exit).  The launch under study ends a cold-cache hipGraph chain of "layers" (producer GEMM with
residual + sums of squares -> norm GEMM), like in the decode layer; shapes = rank 0 of a 7B TP-8 engine at 7 rows.

    python tools/gemm_stamps.py > profiles/r04_gemm_phase_stamps.json      (builds lib/libtriforce_hip_stamps.so itself)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker():
    import torch
    from triforce_amd import ops
    DEV = "cuda:0"
    M, hid, I = int(os.environ.get("STAMP_ROWS", "7")), 4096, 1376
    L = 16
    g = torch.Generator(device=DEV).manual_seed(0)

    def w(n, k):
        return (torch.randn(n, k, generator=g, device=DEV) * 0.02).to(torch.float16)
    wd = [ops.PackedLinear(w(hid, I)) for _ in range(L)]                 # producer: down-like, writes x and ss
    wgu = [ops.PackedLinear(w(2 * I, hid), split=2) for _ in range(L)]
    wqkv = [ops.PackedLinear(w(3 * 512, hid), rope=(4, 128)) for _ in range(L)]
    ln = torch.ones(hid, dtype=torch.float16, device=DEV)
    x = torch.randn(M, hid, generator=g, device=DEV).to(torch.float16)
    act = torch.randn(M, I, generator=g, device=DEV).to(torch.float16)
    ss = ops.ss_buffer(hid, DEV)
    cos = torch.ones(4096, 128, dtype=torch.float16, device=DEV)
    sin = torch.zeros(4096, 128, dtype=torch.float16, device=DEV)
    pos = torch.arange(M, device=DEV)
    kc = torch.zeros(4, 64, 128, dtype=torch.float16, device=DEV)
    vc = torch.zeros(4, 64, 128, dtype=torch.float16, device=DEV)
    ws = ops._SG_WS[torch.device(DEV)]
    out = {}
    for kind in ("gate_up", "qkv"):
        def chain():
            for i in range(L):
                ops.linear(act, wd[i], resid=x, out=x, ss_out=ss)
                if kind == "gate_up":
                    ops.mlp_act(x, wgu[i], ln=ln, eps=1e-5, ss_in=ss)
                else:
                    ops.qkv_rope(x, wqkv[i], ln, 1e-5, cos, sin, pos, kc, vc, 0, 4, 128, ss_in=ss)
        chain()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            chain()
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        nwg = 86 if kind == "gate_up" else 96
        st = ws[(4 << 20):(4 << 20) + nwg * 16 * 8].view(torch.int64).view(nwg, 16).cpu()
        wall0, wall1 = st[:, 8], st[:, 9]
        t0 = int(wall0.min())
        tick_ns = 10.0
        dur_us = (int(wall1.max()) - t0) * tick_ns / 1e3
        cyc_per_us = float(((st[:, 6] - st[:, 0]).double() / ((wall1 - wall0).double() * tick_ns / 1e3)).median())
        names = ["entry->prologue loads issued", "->norm scale known (partials folded)", "->first k-batch done",
                 "->K loop done", "->waves merged (LDS + barrier)", "->exit (epilogue stores issued)"]
        ph = (st[:, 1:7] - st[:, 0:6]).double() / cyc_per_us
        out[kind] = {"workgroups": nwg, "rows": M,
                     "kernel_span_us_first_entry_to_last_exit": round(dur_us, 2),
                     "entry_skew_us_max": round((int(wall0.max()) - t0) * tick_ns / 1e3, 2),
                     "workgroup_lifetime_us_median": round(float(((wall1 - wall0).double() * tick_ns / 1e3).median()), 2),
                     "shader_clock_GHz": round(cyc_per_us / 1e3, 3),
                     "phases_us_median_over_workgroups": {n: round(float(ph[:, i].median()), 2) for i, n in enumerate(names)},
                     "phases_us_max_over_workgroups": {n: round(float(ph[:, i].max()), 2) for i, n in enumerate(names)}}
    print(json.dumps({"source": "python tools/gemm_stamps.py on one MI355X: wave 0 of every workgroup, s_memtime phase stamps "
                                "(instrumented build -DSG_STAMPS=1); the launch under study ends a 16-layer cold-cache hipGraph "
                                "chain (producer GEMM with residual + sums of squares -> norm GEMM); 7B TP-8 rank shapes",
                      "kernels": out}, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker()
    else:
        from triforce_amd.build import LIB_DIR, build_variant
        lib = os.path.join(LIB_DIR, "libtriforce_hip_stamps.so")
        if not os.path.exists(lib):
            build_variant("stamps", ["SG_STAMPS=1"], verbose=False)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker"], env=dict(os.environ, TRIFORCE_HIP_LIB=lib),
                           capture_output=True, text=True)
        sys.stdout.write(r.stdout)
        if r.returncode:
            sys.stderr.write(r.stderr[-2000:])
            sys.exit(1)
