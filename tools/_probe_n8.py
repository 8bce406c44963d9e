import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from triforce_amd import ops
from oracle import ref_ops as R
DEV = "cuda:0"
def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).half()
for M in (1, 7):
    H, D, K = 4, 128, 4096
    N = 3 * H * D
    eps = 1e-5
    x = rnd(M, K, seed=500 + M); ln = (1 + 0.1 * rnd(K, seed=501).float()).half(); w = rnd(N, K, seed=502, scale=0.05)
    cos, sin = R.rope_tables_yarn(D, 4096, 16.0, 256)
    pos = torch.randint(0, 4096, (M,), generator=torch.Generator().manual_seed(M))
    xd, lnd, cd, sd, pd = x.to(DEV), ln.to(DEV), cos.to(DEV), sin.to(DEV), pos.to(DEV)
    pl8 = ops.PackedLinear(w.to(DEV), rope=(H, D))
    ops.N8_ENABLED = False
    pl16 = ops.PackedLinear(w.to(DEV), rope=(H, D))
    ops.N8_ENABLED = True
    def run(pl):
        k = torch.zeros(H, 64, D, dtype=torch.float16, device=DEV); v = torch.zeros_like(k)
        q = ops.qkv_rope(xd, pl, lnd, eps, cd, sd, pd, k, v, 3, H, D)
        return v[:, 3:3 + M].permute(1, 0, 2).contiguous().float().cpu()
    v8, v16 = run(pl8), run(pl16)
    h = R.rms_norm(x, ln, eps)
    wv = R.linear(h, w)[:, 2 * H * D:].view(M, H, D).float()
    truth = (h.double() @ w.double().t())[:, 2 * H * D:].view(M, H, D)
    # truth with exact normalisation too
    xf = x.double(); inv = 1.0 / torch.sqrt((xf * xf).mean(-1, keepdim=True) + eps)
    d = (v8 - v16).abs()
    idx = torch.nonzero(d > 2.5e-3)
    print("M", M, "n8 vs 16:", "max", float(d.max()), "n>2.5e-3", len(idx), "| n8 vs oracle max", float((v8 - wv).abs().max()),
          "| 16 vs oracle max", float((v16 - wv).abs().max()), "| n8 vs truth", float((v8.double() - truth).abs().max()), "16 vs truth", float((v16.double() - truth).abs().max()))
    for i in idx[:6]:
        i = tuple(i.tolist())
        print("   ", i, "n8", float(v8[i]), "w16", float(v16[i]), "oracle", float(wv[i]), "truth(h fp16)", float(truth[i]))
