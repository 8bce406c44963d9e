#!/bin/bash
# one gpurun call: tests, bench, split sweep over the attention variants, glue microbench, rocprof kernel trace
O=gpurun_out/r2b; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -15 $O/pytest.log
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 2500 $O/bench.json
for v in default occ3 q2occ2 q2occ2lv; do
  if [ $v = default ]; then unset TRIFORCE_HIP_LIB; else export TRIFORCE_HIP_LIB=$PWD/triforce_amd/lib/libtriforce_hip_$v.so; fi
  timeout 300 python tools/nsplit_sweep.py $v > $O/sweep_$v.log 2>&1; echo "sweep $v rc=$?"
done
unset TRIFORCE_HIP_LIB
timeout 200 python tools/microbench.py > $O/microbench.log 2>&1; cp gpurun_out/microbench.json $O/ 2>/dev/null; grep -E "^glue" $O/microbench.log | cut -c1-1200
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-steps 0 > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_prof.err; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT; find $O/prof -name "*kernel_stats.csv" | head -2; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" | cut -c1-160
find $O/prof -name "*kernel_trace.csv" -size +60M -delete
