#!/bin/bash
# round 4, GPU call 24: HIP runtime queue / dispatch switches against the driver-form bench line (no profiler)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04c24
mkdir -p $O
run() {
  tag=$1; shift
  env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-steps 0 2>$O/err_$tag.txt | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'env': '$*', 'tokens_per_s': d['value'], 'ms_per_step': d['ms_per_step'], 'step_overhead_us': d['step_overhead_us'], 'stage_latency_us': d['stage_latency_us']}))" | tee -a $O/runtime_queue_knobs.jsonl
}
run d1 A=1
run dd0 AMD_DIRECT_DISPATCH=0
run q1 GPU_MAX_HW_QUEUES=1
run q2 GPU_MAX_HW_QUEUES=2
run d2 A=2
tail -n 3 $O/err_dd0.txt $O/err_q1.txt | cut -c1-200
