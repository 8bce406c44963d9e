#!/bin/bash
# round 4, GPU call 15: epilogue operands of the plain GEMMs fetched behind the first weight batch (SG_EPI_LATE) — tests,
# then stage latencies default vs SG_EPI_LATE=0 (single GPU 7B, TP-8 rank shards 7B / 13B)
set -x
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04c15
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/pytest.txt
cat $O/pytest.txt
L=$PWD/triforce_amd/lib
run() {   # tag, lib
  tag=$1; lib=$2
  TRIFORCE_HIP_LIB=$lib timeout 300 python tools/verify_bench.py "$tag" 2>$O/vb_$tag.err | grep '^{' >> $O/verify_bench_variants.jsonl
  TRIFORCE_HIP_LIB=$lib timeout 300 python tools/tp_shard_bench.py llama-7B-128K 8 --gamma 6 --prefill 124928 --budget 4096 --local-exchange --gemm-exchange 2>$O/tp_$tag.err | grep '^{' | sed "s/^{/{\"variant\": \"$tag\", /" >> $O/tp_shard_variants.jsonl
  TRIFORCE_HIP_LIB=$lib timeout 300 python tools/tp_shard_bench.py llama-13B-128K 8 --local-exchange --gemm-exchange 2>>$O/tp13_$tag.err | grep '^{' | sed "s/^{/{\"variant\": \"$tag\", /" >> $O/tp_shard_variants.jsonl
}
run default $L/libtriforce_hip.so
run epilate0 $L/libtriforce_hip_epilate0.so
run default2 $L/libtriforce_hip.so
run epilate0b $L/libtriforce_hip_epilate0.so
python - <<'PY'
import json
for f in ("verify_bench_variants", "tp_shard_variants"):
    for l in open(f"gpurun_out/r04c15/{f}.jsonl"):
        d = json.loads(l); print(f[:8], d.get("tag") or d.get("variant"), d.get("model", ""), {k: v for k, v in d.items() if k.endswith("_us")})
PY
tail -n 2 $O/*.err | cut -c1-200
