#!/bin/bash
# A/B of the narrow-panel (8-row) norm GEMMs on a tensor-parallel rank's shard, one GPU:
# rank 0's shard of a W-way engine, exchanges against a one-rank group, whole-forward hipGraphs.
#   gpurun --timeout 900 -- 'bash tools/ab_narrow_panels.sh'        -> gpurun_out/n8/*.jsonl
O=gpurun_out/n8; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -q -x -k "narrow_panel or split_across or swiglu_norm or qkv_rope_fused" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
X="--local-exchange --gemm-exchange"
run() {  # label, env..., -- args
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python tools/tp_shard_bench.py "$@" 2>>$O/err.log | grep '^{' | sed "s/^{/{\"variant\": \"$label\", /" >> $O/tp_shard_ab.jsonl
}
rm -f $O/tp_shard_ab.jsonl
for rep in 1 2; do
  run n8_off TRIFORCE_GEMM_N8=0 -- llama-7B-128K 8 --gamma 6 --prefill 124928 --budget 4096 $X
  run n8_on  TRIFORCE_GEMM_N8=1 -- llama-7B-128K 8 --gamma 6 --prefill 124928 --budget 4096 $X
done
run n8_off TRIFORCE_GEMM_N8=0 -- llama-13B-128K 8 --gamma 16 --prefill 130048 --budget 12288 $X
run n8_on  TRIFORCE_GEMM_N8=1 -- llama-13B-128K 8 --gamma 16 --prefill 130048 --budget 12288 $X
run n8_off TRIFORCE_GEMM_N8=0 -- llama-7B-128K 4 --gamma 6 --prefill 124928 --budget 4096 $X
run n8_on_200 TRIFORCE_GEMM_N8=1 TRIFORCE_GEMM_N8_MAX_PANELS=200 -- llama-7B-128K 4 --gamma 6 --prefill 124928 --budget 4096 $X
python - <<'PY'
import json
for l in open("gpurun_out/n8/tp_shard_ab.jsonl"):
    j = json.loads(l)
    print(j["variant"], j["target"], j["emulated_world"], j["gamma"], "draft", j["draft_step_us"], "rv", j["retrieval_verify_us"], "tv", j["target_verify_us"], "ar", j["ar_step_eager_us"])
PY
