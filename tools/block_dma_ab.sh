#!/bin/bash
O=gpurun_out/block_dma; mkdir -p $O
for L in default dma1 dma0; do
  LIB=$PWD/triforce_amd/lib/libtriforce_hip.so; [ $L != default ] && LIB=$PWD/triforce_amd/lib/libtriforce_hip_$L.so
  TRIFORCE_HIP_LIB=$LIB python - "$L" <<'PY'
import json, os, sys, torch
sys.path.insert(0, os.getcwd())
from triforce_amd import ops
DEV = "cuda:0"
tag = sys.argv[1]
def timeit(fns, iters):
    for f in fns: f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(iters): fns[i % len(fns)]()
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / iters * 1e3)
    return min(ts)
g = torch.Generator(device=DEV).manual_seed(0)
T, P, H = 512, 124928, 32
kvs = [(torch.randn(H, P + T, 128, generator=g, device=DEV, dtype=torch.float16), torch.randn(H, P + T, 128, generator=g, device=DEV, dtype=torch.float16)) for _ in range(2)]
q = torch.randn(T, H, 128, generator=g, device=DEV, dtype=torch.float16)
vis = torch.tril(torch.rand(T, T, generator=g, device=DEV) < 0.2) | torch.eye(T, dtype=torch.bool, device=DEV)
bits = ops.pack_tree_mask(vis)
res = {"lib": tag}
res["attn_tree_verify_512_us"] = round(timeit([(lambda kv=kv: ops.attn_tree(q, kv[0], kv[1], P + T, 0.08838834764831845, bits, P)) for kv in kvs], 8), 1)
out_tree = ops.attn_tree(q, kvs[0][0], kvs[0][1], P + T, 0.08838834764831845, bits, P)
res["tree_checksum"] = float(out_tree.float().abs().sum())
q128 = q[:128].contiguous()
res["attn_block_128rows_us"] = round(timeit([(lambda kv=kv: ops.attn_block(q128, kv[0], kv[1], P, 0.08837890625)) for kv in kvs], 8), 1)
res["block_checksum"] = float(ops.attn_block(q128, kvs[0][0], kvs[0][1], P, 0.08837890625).float().abs().sum())
print(json.dumps(res), flush=True)
PY
done > $O/block_dma_ab.jsonl 2> $O/block_dma_ab.err; cat $O/block_dma_ab.jsonl; tail -2 $O/block_dma_ab.err
TRIFORCE_HIP_LIB=$PWD/triforce_amd/lib/libtriforce_hip_dma0.so python -m pytest tests/test_gpu_ops.py tests/test_gpu_sequoia.py -q -k "prefill or block or tree or sequoia or Sequoia" > $O/pytest_dma0.log 2>&1; echo "pytest dma0 rc=$?"; tail -3 $O/pytest_dma0.log
