#!/bin/bash
# round 4, GPU call 13: kernel arguments preloaded into SGPRs (-amdgpu-kernarg-preload-count=14 + argument order) and the
# norm-GEMM prologue as one basic block (partials first, fold under the weights in flight): GEMM / layer tests, then stage
# latencies of the single-GPU forwards and of a TP-8 rank's shard for the default build and three variants
set -x
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04c13
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_layouts.py tests/test_gpu_e2e.py -x -q 2>&1 | tail -6 > $O/pytest.txt
cat $O/pytest.txt
L=$PWD/triforce_amd/lib
run() {   # tag, lib
  tag=$1; lib=$2
  TRIFORCE_HIP_LIB=$lib timeout 300 python tools/verify_bench.py "$tag" 2>$O/vb_$tag.err | grep '^{' >> $O/verify_bench_variants.jsonl
  TRIFORCE_HIP_LIB=$lib timeout 300 python tools/tp_shard_bench.py llama-7B-128K 8 --gamma 6 --prefill 124928 --budget 4096 --local-exchange --gemm-exchange 2>$O/tp_$tag.err | grep '^{' | sed "s/^{/{\"variant\": \"$tag\", /" >> $O/tp_shard_variants.jsonl
}
run default $L/libtriforce_hip.so
run nopreload $L/libtriforce_hip_nopreload.so
run prol1 $L/libtriforce_hip_prol1.so
run lnpre $L/libtriforce_hip_lnpre.so
run default2 $L/libtriforce_hip.so
TRIFORCE_HIP_LIB=$L/libtriforce_hip.so timeout 300 python tools/tp_shard_bench.py llama-13B-128K 8 --local-exchange --gemm-exchange 2>>$O/tp13.err | grep '^{' | sed "s/^{/{\"variant\": \"default\", /" >> $O/tp_shard_variants.jsonl
TRIFORCE_HIP_LIB=$L/libtriforce_hip_nopreload.so timeout 300 python tools/tp_shard_bench.py llama-13B-128K 8 --local-exchange --gemm-exchange 2>>$O/tp13.err | grep '^{' | sed "s/^{/{\"variant\": \"nopreload\", /" >> $O/tp_shard_variants.jsonl
python - <<'PY'
import json
for f in ("verify_bench_variants", "tp_shard_variants"):
    for l in open(f"gpurun_out/r04c13/{f}.jsonl"):
        d = json.loads(l); print(f[:8], d.get("tag") or d.get("variant"), d.get("model", ""), {k: v for k, v in d.items() if k.endswith("_us")})
PY
tail -n 2 $O/*.err | cut -c1-200
