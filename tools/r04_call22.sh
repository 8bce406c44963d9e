#!/bin/bash
# round 4, GPU call 22: the record poll with / without sched_yield(), round-4 host path, idle time under rocprofv3 — alternating
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R"
O=gpurun_out/r04c22
mkdir -p $O
nproc; cat /proc/cpuinfo | grep "model name" | head -1
cd /tmp && export TMPDIR=/tmp
for tag in y1a y0a y1b y0b leg; do
  case $tag in y1*) Y=1; F=1;; y0*) Y=0; F=1;; leg) Y=0; F=0;; esac
  TRIFORCE_POLL_YIELD=$Y TRIFORCE_HOST_FAST=$F timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof_$tag -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-steps 0 > $R/$O/bench_$tag.json 2> $R/$O/bench_$tag.err
  T=$(ls -S $R/$O/prof_$tag/*/*kernel_trace.csv | head -1)
  python $R/tools/gap_analysis.py $T --steps 19 > $R/$O/gap_$tag.txt 2>&1
  rm -rf $R/$O/prof_$tag
done
cd $R
python - <<'PY'
import json
for t in ("y1a", "y0a", "y1b", "y0b", "leg"):
    try:
        d = json.load(open(f"gpurun_out/r04c22/bench_{t}.json"))
        g = open(f"gpurun_out/r04c22/gap_{t}.txt").read().splitlines()
        hop = [l for l in g if "middle_accept_kernel" in l and "draft_embed" in l][:1]
        print(t, d["value"], d["ms_per_step"], g[1], hop)
    except Exception as e:
        print(t, "failed", e)
PY
