#!/bin/bash
# One gpurun call that reproduces the round's evidence: GPU tests, smoke, the driver-form bench line, and the
# rocprofv3 kernel-trace summary of the same command.  Outputs under gpurun_out/validate/.
#   gpurun --timeout 1500 -- 'bash tools/gpu_validate.sh'
O=gpurun_out/validate; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log; grep "\[parity\]" $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 800 $O/bench.json
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-steps 0 > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err; echo "rocprof rc=$?"
cd $R; find $O/prof -name "*kernel_trace.csv" -size +60M -delete
