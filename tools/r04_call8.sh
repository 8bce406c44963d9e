#!/bin/bash
# round 4, GPU call 8: twice the weights in flight per wave on few-panel grids (UX = 2) — TP-shard stage latencies A/B, phase
# stamps, GEMM / exchange tests
set -x
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04c8
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_layouts.py tests/test_gpu_tp_offload.py -x -q -k "gemm or skinny or swiglu or qkv or layouts or exchange or down_proj" 2>&1 | tail -8 > $O/pytest_gemm.txt
cat $O/pytest_gemm.txt
for cfg in "7Bw8:llama-7B-128K 8 --gamma 6 --prefill 124928 --budget 4096" "7Bw4:llama-7B-128K 4 --gamma 6 --prefill 124928 --budget 4096" "13Bw8:llama-13B-128K 8" "7Bw2g16:llama-7B-128K 2"; do
  tag=${cfg%%:*}; a=${cfg#*:}
  timeout 600 python tools/tp_shard_bench.py $a --local-exchange --gemm-exchange 2>$O/tp_${tag}.err | grep '^{' | sed "s/^{/{\"variant\": \"UX = 2 on grids of <= 200 panels and in the fused exchange\", /" >> $O/tp_shard.jsonl
  TRIFORCE_GEMM_DEEP_PANELS=0 timeout 600 python tools/tp_shard_bench.py $a --local-exchange --gemm-exchange 2>$O/tp_${tag}_nodeep.err | grep '^{' | sed "s/^{/{\"variant\": \"TRIFORCE_GEMM_DEEP_PANELS=0 (UX = 2 only in the fused exchange)\", /" >> $O/tp_shard.jsonl
done
timeout 300 python tools/gemm_stamps.py > $O/gemm_phase_stamps_7rows.json 2> $O/stamps7.err
TRIFORCE_GEMM_DEEP_PANELS=0 timeout 300 python tools/gemm_stamps.py > $O/gemm_phase_stamps_7rows_nodeep.json 2>> $O/stamps7.err
cat $O/gemm_phase_stamps_7rows.json | head -60
for f in $O/*.err; do echo "== $f"; tail -n 2 $f; done
