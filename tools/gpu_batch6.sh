#!/bin/bash
O=gpurun_out/r2f; mkdir -p $O
python -m pytest tests/test_gpu_e2e.py -m gpu -q -k "stochastic or 7b_dimension or near_ties" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log; grep "\[parity\]" $O/pytest.log
TRIFORCE_PREFILL_CHUNK=2048 timeout 900 python bench.py --prefill 130048 --budget 12288 --gamma 16 --on-chip 9 --steps 8 --warmup 2 --no-cpu-baseline --random-steps 0 > $O/bench_offload.json 2> $O/bench_offload.err; echo "offload rc=$?"; tail -c 3000 $O/bench_offload.json; tail -5 $O/bench_offload.err
