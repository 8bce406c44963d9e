bash tools/ab_inner_loop.sh 1
X="--local-exchange --gemm-exchange"
bash tools/profile_tp_shard.sh tp8_7b_n8 TRIFORCE_GEMM_N8=1 -- llama-7B-128K 8 --gamma 6 --prefill 124928 --budget 4096 $X | head -30
mkdir -p gpurun_out/n8
for v in "u5:TRIFORCE_GEMM_N8_U=5" "w4:TRIFORCE_HIP_LIB=$PWD/triforce_amd/lib/libtriforce_hip_n8w4.so" "base:TRIFORCE_GEMM_N8=1"; do
  label=${v%%:*}; envs=${v#*:}
  for rep in 1 2; do
  env $envs python tools/tp_shard_bench.py llama-7B-128K 8 --gamma 6 --prefill 124928 --budget 4096 $X 2>/dev/null | grep '^{' | sed "s/^{/{\"variant\": \"$label\", /" >> gpurun_out/n8/variants.jsonl
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/n8/variants.jsonl"):
    j = json.loads(l); print(j["variant"], "rv", j["retrieval_verify_us"], "tv", j["target_verify_us"], "draft", j["draft_step_us"])
PY
