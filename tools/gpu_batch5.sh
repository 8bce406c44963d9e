#!/bin/bash
O=gpurun_out/r2e; mkdir -p $O
free -g | head -2
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 2600 $O/bench.json; tail -3 $O/bench.err
python bench.py --target lwm-128K --steps 20 --warmup 5 --random-steps 0 --no-cpu-baseline > $O/bench_lwm.json 2> $O/bench_lwm.err; echo "lwm rc=$?"; tail -c 1800 $O/bench_lwm.json; tail -3 $O/bench_lwm.err
TRIFORCE_PREFILL_CHUNK=2048 timeout 600 python bench.py --prefill 130048 --budget 12288 --gamma 16 --on-chip 9 --steps 8 --warmup 2 --no-cpu-baseline --random-steps 0 > $O/bench_offload.json 2> $O/bench_offload.err; echo "offload rc=$?"; tail -c 2600 $O/bench_offload.json; tail -5 $O/bench_offload.err
