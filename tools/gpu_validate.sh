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
# HBM traffic of the roofline kernel (two separate --pmc passes, kernel-trace only) -> the figure bench.py quotes as roofline.traffic
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -- python $R/tools/pmc_attn.py > $R/$O/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -- python $R/tools/pmc_attn.py > $R/$O/pmc_write.log 2>&1; echo "pmc write rc=$?"
cd $R; python tools/pmc_reduce.py $O/pmc_fetch $O/pmc_write $O/pmc_attn_target_verify.json "tools/gpu_validate.sh"
find $O/prof $O/pmc_fetch $O/pmc_write -name "*kernel_trace.csv" -size +20M -delete
# the other full-size configs (BASELINE configs[2], and configs[3] at world size 1: offloading tier), only with "all"
if [ "$1" = "all" ]; then
  python bench.py --target lwm-128K --steps 20 --warmup 5 --no-cpu-baseline --random-steps 0 > $O/bench_lwm.json 2> $O/bench_lwm.err; echo "lwm rc=$?"; tail -c 300 $O/bench_lwm.json
  TRIFORCE_PREFILL_CHUNK=2048 python bench.py --prefill 130048 --budget 12288 --gamma 16 --on-chip 9 --steps 8 --warmup 2 --no-cpu-baseline --random-steps 0 > $O/bench_offload.json 2> $O/bench_offload.err; echo "offload rc=$?"; tail -c 300 $O/bench_offload.json
  python tools/verify_bench.py final > $O/verify_bench.json 2> $O/verify_bench.err; echo "verify_bench rc=$?"; tail -c 400 $O/verify_bench.json
fi
