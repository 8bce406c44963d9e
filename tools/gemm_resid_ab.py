"""Cold-cache timing of the residual-epilogue GEMMs of the decode layer (o_proj 4096 x 4096, down_proj 4096 x 11008,
7 rows, residual add in place + sum-of-squares hand-off) for the shipped library and, when built, the variant that
loads the residual operands at the tail (python tools/ab_variants.py reslate).  Graph chains over rotating weight copies.
    gpurun -- 'python tools/gemm_resid_ab.py'"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(tag):
    import torch
    from triforce_amd import ops
    dev = "cuda:0"
    out = {"tag": tag}
    for name, N, K in (("o_proj", 4096, 4096), ("down_proj", 4096, 11008)):
        copies = max(2, int(700e6 // (N * K * 2)) + 1)
        pls = [ops.PackedLinear(torch.randn(N, K, device=dev, dtype=torch.float16) * 0.02) for _ in range(copies)]
        for M in (7, 17):
            x = torch.randn(M, K, device=dev, dtype=torch.float16)
            res = torch.randn(M, N, device=dev, dtype=torch.float16)
            ss = ops.ss_buffer(N, dev)
            n = 48
            fns = [(lambda p=p: ops.linear(x, p, resid=res, out=res, ss_out=ss)) for p in pls]
            for f in fns:
                f()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(n):
                    fns[i % len(fns)]()
            g.replay()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                g.replay()
                e.record()
                torch.cuda.synchronize()
                best = min(best, s.elapsed_time(e) / n * 1e3)
            out[f"{name}_rows{M}_us"] = round(best, 2)
        del pls
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    from triforce_amd.build import LIB_PATH
    libs = {"default": LIB_PATH}
    alt = LIB_PATH.replace(".so", "_reslate.so")
    if os.path.exists(alt):
        libs["reslate"] = alt
    for rep in range(2):
        for tag, lib in libs.items():
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", tag],
                               env=dict(os.environ, TRIFORCE_HIP_LIB=lib), capture_output=True, text=True)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print(lines[-1] if lines else json.dumps({"tag": tag, "error": r.stderr[-400:]}), flush=True)
