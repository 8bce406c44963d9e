#!/bin/bash
# round 4, GPU call 23: round-3 / round-4 host path WITHOUT the profiler attached (tracing costs host time per launch):
# driver-form bench, alternating, three each
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04c23
mkdir -p $O
for i in 1 2 3; do
  for F in 1 0; do
    TRIFORCE_HOST_FAST=$F python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-steps 0 2>$O/err_${F}_$i.txt | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'host_fast': $F, 'rep': $i, 'tokens_per_s': d['value'], 'ms_per_step': d['ms_per_step'], 'step_overhead_us': d['step_overhead_us'], 'stage_latency_us': d['stage_latency_us']}))" | tee -a $O/host_path_plain.jsonl
  done
done
