#!/bin/bash
# round 4, GPU call 12: HIP runtime launch-path switches (kernel arguments in device memory, graph packet capture, kernarg
# copy path) against the stage latencies of the single-GPU forwards and of a TP-8 rank's shard (the launch-bound one)
set -x
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04c12
mkdir -p $O
run() {   # tag, env assignments...
  tag=$1; shift
  env "$@" timeout 300 python tools/verify_bench.py "$tag" 2>$O/vb_$tag.err | grep '^{' | sed "s/^{/{\"env\": \"$*\", /" >> $O/verify_bench_env.jsonl
  env "$@" timeout 300 python tools/tp_shard_bench.py llama-7B-128K 8 --gamma 6 --prefill 124928 --budget 4096 --local-exchange --gemm-exchange 2>$O/tp_$tag.err | grep '^{' | sed "s/^{/{\"env\": \"$*\", /" >> $O/tp_shard_env.jsonl
}
run default A=1
run devkernarg1 HIP_FORCE_DEV_KERNARG=1
run devkernarg0 HIP_FORCE_DEV_KERNARG=0
run nopacketcapture DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run kernargcopy0 DEBUG_HIP_KERNARG_COPY_OPT=0
run hdpflushwa0 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0
run default2 A=2
python - <<'PY'
import json
for f in ("verify_bench_env", "tp_shard_env"):
    for l in open(f"gpurun_out/r04c12/{f}.jsonl"):
        d = json.loads(l); print(f, d.get("env"), {k: v for k, v in d.items() if k.endswith("_us")})
PY
tail -n 2 $O/*.err | cut -c1-200
