#!/bin/bash
# A/B of the round-5 decode loop on one GPU, same box, alternating: the shipped loop (one hipGraph per inner
# iteration, launch plans, step set-up on the device) against each piece switched off, down to rounds 2-4's loop — the driver-form
# bench line per variant, plus the parity tests that pin the token streams.   gpurun --timeout 1500 -- 'bash tools/ab_inner_loop.sh [reps]'
O=gpurun_out/inner; mkdir -p $O; REPS=${1:-1}
python -m pytest tests/test_gpu_ops.py -q -x -k "narrow_panel or cursor_forms or accept or sample" > $O/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -2 $O/pytest_ops.log
python -m pytest tests/test_gpu_e2e.py -q -x > $O/pytest_e2e.log 2>&1; echo "e2e rc=$?"; tail -3 $O/pytest_e2e.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-steps 0"
rm -f $O/bench_ab.jsonl
for rep in $(seq $REPS); do
  for v in "shipped:TRIFORCE_INNER_GRAPH=1" "host-setup:TRIFORCE_STEP_ON_DEVICE=0" "no-plans:TRIFORCE_STEP_ON_DEVICE=0 TRIFORCE_HOST_PLANS=0" \
           "round4-loop:TRIFORCE_INNER_GRAPH=0 TRIFORCE_HOST_PLANS=0"; do
    label=${v%%:*}; envs=${v#*:}
    env $envs $B 2>>$O/bench.err | grep '^{' | sed "s/^{/{\"variant\": \"$label\", /" >> $O/bench_ab.jsonl
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/inner/bench_ab.jsonl"):
    j = json.loads(l)
    s = j["stage_latency_us"]
    print(j["variant"], "tok/s", j["value"], "ms/step", j["ms_per_step"], "overhead_us", j["step_overhead_us"], "tokens/step", j["tokens_per_step"],
          "rv", s.get("retrieval_verify_us"), "tv", s.get("target_verify_us"), "draft", s.get("draft_step_us"))
PY
