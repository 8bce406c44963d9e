#!/bin/bash
# round 4, GPU call 17: split-KV attention kernels with every prologue argument preloaded (one scalar wait — the device key
# count — instead of two dependent ones): attention + end-to-end tests, stage latencies against the previous build
set -x
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04c17
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_layouts.py tests/test_gpu_e2e.py tests/test_gpu_configs.py -x -q > $O/pytest_full.txt 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $O/pytest_full.txt
L=$PWD/triforce_amd/lib
run() {   # tag, lib
  tag=$1; lib=$2
  TRIFORCE_HIP_LIB=$lib timeout 300 python tools/verify_bench.py "$tag" 2>$O/vb_$tag.err | grep '^{' >> $O/verify_bench_variants.jsonl
  TRIFORCE_HIP_LIB=$lib timeout 300 python tools/tp_shard_bench.py llama-7B-128K 8 --gamma 6 --prefill 124928 --budget 4096 --local-exchange --gemm-exchange 2>$O/tp_$tag.err | grep '^{' | sed "s/^{/{\"variant\": \"$tag\", /" >> $O/tp_shard_variants.jsonl
}
run default $L/libtriforce_hip.so
run prev $L/libtriforce_hip_prev.so
run default2 $L/libtriforce_hip.so
run prev2 $L/libtriforce_hip_prev.so
python - <<'PY'
import json
for f in ("verify_bench_variants", "tp_shard_variants"):
    for l in open(f"gpurun_out/r04c17/{f}.jsonl"):
        d = json.loads(l); print(f[:8], d.get("tag") or d.get("variant"), d.get("model", ""), {k: v for k, v in d.items() if k.endswith("_us")})
PY
tail -n 2 $O/*.err | cut -c1-200
