#!/bin/bash
# round 4, GPU call 26: the last host -> device copies of the decode loop (token lists of the eager-sampled verify steps)
# replaced by kernel arguments: driver-form bench without a profiler, alternating (mask 31 = all, 15 = round 4 so far)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04c26
mkdir -p $O
for i in 1 2; do
  for M in 31 15; do
    TRIFORCE_HOST_FAST_MASK=$M python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-steps 0 2>$O/err_${M}_$i.txt | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'mask': $M, 'rep': $i, 'tokens_per_s': d['value'], 'ms_per_step': d['ms_per_step'], 'step_overhead_us': d['step_overhead_us']}))" | tee -a $O/host_path_mask31.jsonl
  done
done
