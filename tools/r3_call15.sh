#!/bin/bash
O=gpurun_out/r3k; mkdir -p $O
python -m pytest tests/test_gpu_e2e.py tests/test_gpu_ops.py -q -k "draft_prefill_graph or stale_partials or static_verify" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --random-steps 0 > $O/bench.json 2> $O/bench.err; python -c "
import json; j=json.load(open('$O/bench.json')); print({k: j[k] for k in ('value','ms_per_step','prefill_seconds','setup_seconds')})"
TRIFORCE_DRAFT_PREFILL_GRAPH=0 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --random-steps 0 > $O/bench_nograph.json 2> $O/bench_nograph.err; python -c "
import json; j=json.load(open('$O/bench_nograph.json')); print('no draft-prefill graph', {k: j[k] for k in ('value','ms_per_step','prefill_seconds')})"
