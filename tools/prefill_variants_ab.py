"""A/B of prefill-attention (tf_attn_prefill) library variants at 7B shapes: 32 heads x 128, one 1024-row chunk at the end
of 32 768 / 124 928 cached keys.  TF/s = causal QK^T + PV flops / time.  Also the largest deviation from attention
accumulated in fp64 on the device (a 64-row slice), so a variant that changes the arithmetic shows by how much.
    python tools/prefill_variants_ab.py <variant> [<variant> ...]      (variants: tools/ab_variants.py; "default" = shipped)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(tag):
    import torch
    from triforce_amd import ops
    DEV = torch.device("cuda", 0)
    H, D, chunk = 32, 128, 1024
    g = torch.Generator(device=DEV).manual_seed(1)
    cap = 124928
    k = torch.randn(H, cap, D, generator=g, device=DEV, dtype=torch.float16)
    v = torch.randn(H, cap, D, generator=g, device=DEV, dtype=torch.float16)
    q = torch.randn(chunk, H, D, generator=g, device=DEV, dtype=torch.float16)
    scale = 0.08837890625
    for sk in (32768, 124928):
        flops = 4.0 * H * D * (chunk * (sk - chunk) + chunk * (chunk + 1) / 2)
        out = ops.attn_prefill(q, k, v, sk, scale)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                ops.attn_prefill(q, k, v, sk, scale)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 4)
        us = min(ts)
        row = {"lib": tag, "chunk_rows": chunk, "sk": sk, "us": round(us, 1), "TFps": round(flops / us / 1e6, 1)}
        if sk == 32768:                                            # exactness on 2 heads x 64 rows (fp64 on the device)
            r0, hs = chunk - 64, slice(0, 2)
            s = torch.einsum("qhd,hkd->hqk", q[r0:, hs].double(), k[hs, :sk].double()) * scale
            qi = torch.arange(r0, chunk, device=DEV).view(-1, 1)
            kj = torch.arange(sk, device=DEV).view(1, -1)
            s = s.masked_fill(kj > (sk - chunk + qi), float("-inf"))
            want = torch.einsum("hqk,hkd->qhd", torch.softmax(s, dim=-1), v[hs, :sk].double())
            got = out.view(chunk, H, D)[r0:, hs].double()
            row["max_abs_err_vs_fp64"] = float((got - want).abs().max())
            row["mean_abs_err_vs_fp64"] = float((got - want).abs().mean())
            # how many fp16 outputs are NOT the correctly rounded value of the fp64 result (round 3's measure for the
            # decode kernel: 36 % with P rounded once to fp16, 0.16 % with P fed as hi + lo)
            row["frac_not_correctly_rounded"] = float((got.half() != want.half()).float().mean())
        print(json.dumps(row), flush=True)
    # Sequoia verify: 512 tree rows behind a 124 928-token prefix (TREE form of the LDS block kernel), random ancestor masks
    sk, T = 124928 - 512, 512
    vis = torch.tril(torch.rand(T, T, generator=g, device=DEV) < 0.1) | torch.eye(T, dtype=torch.bool, device=DEV)
    bits = ops.pack_tree_mask(vis)
    qt = torch.randn(T, H, D, generator=g, device=DEV, dtype=torch.float16)
    ops.attn_tree(qt, k, v, sk + T, scale, bits, sk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        ops.attn_tree(qt, k, v, sk + T, scale, bits, sk)
    e1.record()
    torch.cuda.synchronize()
    print(json.dumps({"lib": tag, "tree_verify_512_nodes_us": round(e0.elapsed_time(e1) * 1e3 / 4, 1)}), flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--worker":
        return worker(sys.argv[2])
    from triforce_amd.build import LIB_DIR, LIB_PATH
    for n in sys.argv[1:]:
        lib = LIB_PATH if n == "default" else os.path.join(LIB_DIR, f"libtriforce_hip_{n}.so")
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", n], env=dict(os.environ, TRIFORCE_HIP_LIB=lib),
                             capture_output=True, text=True)
        sys.stdout.write("".join(l + "\n" for l in out.stdout.splitlines() if l.startswith("{")))
        if out.returncode:
            print(json.dumps({"lib": n, "failed": out.stderr[-500:]}))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
