#!/bin/bash
# round 4, GPU call 20: the decode step's host path, round-3 form against round-4 form ON ONE BOX (TRIFORCE_HOST_FAST=0 / 1):
# idle time between kernels of 19 steps of the driver-form bench under rocprofv3, two alternations
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R"
O=gpurun_out/r04c20
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for tag in fast1 legacy1 fast2 legacy2; do
  case $tag in fast*) F=1;; *) F=0;; esac
  TRIFORCE_HOST_FAST=$F timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof_$tag -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-steps 0 > $R/$O/bench_$tag.json 2> $R/$O/bench_$tag.err
  T=$(ls -S $R/$O/prof_$tag/*/*kernel_trace.csv | head -1)
  python $R/tools/gap_analysis.py $T --steps 19 > $R/$O/gap_$tag.txt 2>&1
  head -12 $R/$O/gap_$tag.txt | cut -c1-150
  rm -rf $R/$O/prof_$tag
done
cd $R
python - <<'PY'
import json
for t in ("fast1", "legacy1", "fast2", "legacy2"):
    d = json.load(open(f"gpurun_out/r04c20/bench_{t}.json"))
    print(t, d["value"], d["ms_per_step"], d["step_overhead_us"], open(f"gpurun_out/r04c20/gap_{t}.txt").read().splitlines()[1])
PY
