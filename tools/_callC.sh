O=gpurun_out/inner; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -q -x -k "narrow_panel or cursor_forms or accept or sample" > $O/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -2 $O/pytest_ops.log
python -m pytest tests/test_gpu_e2e.py -q -x -k "one_launch_inner or static_verify or full_scale or golden" > $O/pytest_e2e.log 2>&1; echo "e2e rc=$?"; tail -3 $O/pytest_e2e.log
rm -f $O/hop_ab.jsonl
for rep in 1 2; do
for v in "inner+lanes:TRIFORCE_X=1" "inner:TRIFORCE_LANES=0" "lanes:TRIFORCE_INNER_GRAPH=0" "r04:TRIFORCE_INNER_GRAPH=0 TRIFORCE_LANES=0"; do
  label=${v%%:*}; envs=${v#*:}
  env $envs python tools/hop_trace.py 2>>$O/hop.err | grep '^{' | sed "s/^{/{\"variant\": \"$label\", /" >> $O/hop_ab.jsonl
done
done
python - <<'PY'
import json
for l in open("gpurun_out/inner/hop_ab.jsonl"):
    j = json.loads(l)
    if "stage_latency_us" in j:
        s = j["stage_latency_us"]
        print(j["variant"], "tok/s", j["value"], "ms/step", j["ms_per_step"], "overhead_us", j["step_overhead_us"], "tokens/step", j["tokens_per_step"],
          "rv", s.get("retrieval_verify_us"), "tv", s.get("target_verify_us"), "draft", s.get("draft_step_us"))
    else:
        print(j["variant"], j)
PY
