#!/bin/bash
# round 4, GPU call 21: which piece of the round-4 host path costs idle time on this box — one piece at a time
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R"
O=gpurun_out/r04c21
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for M in 15 0 1 2 4 8; do
  TRIFORCE_HOST_FAST_MASK=$M TRIFORCE_HOST_FAST=$([ $M = 0 ] && echo 0 || echo 1) timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof_$M -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-steps 0 > $R/$O/bench_$M.json 2> $R/$O/bench_$M.err
  T=$(ls -S $R/$O/prof_$M/*/*kernel_trace.csv | head -1)
  python $R/tools/gap_analysis.py $T --steps 19 > $R/$O/gap_$M.txt 2>&1
  rm -rf $R/$O/prof_$M
done
cd $R
python - <<'PY'
import json
for t in (15, 0, 1, 2, 4, 8):
    try:
        d = json.load(open(f"gpurun_out/r04c21/bench_{t}.json"))
        print(t, d["value"], d["ms_per_step"], open(f"gpurun_out/r04c21/gap_{t}.txt").read().splitlines()[1])
    except Exception as e:
        print(t, "failed", e)
PY
tail -n 3 $O/bench_15.err
