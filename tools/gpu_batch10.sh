#!/bin/bash
O=gpurun_out/r2j; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log; grep "\[parity\]" $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1500 $O/bench.json
python bench.py --steps 12 --warmup 3 --eager-comparator --random-steps 0 --no-cpu-baseline > $O/bench_eager.json 2> $O/bench_eager.err; echo "eager rc=$?"; python - <<'PY'
import json
j=json.loads(open("gpurun_out/r2j/bench_eager.json").read().strip().splitlines()[-1])
print(json.dumps(j.get("eager_torch_comparator"))[:1500]); print(j["value"], j["ar_baseline_tokens_per_s"])
PY
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-steps 0 > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err; echo "rocprof rc=$?"
cd $R; find $O/prof -name "*kernel_trace.csv" -size +70M -delete; ls -la $O/prof/*/ | head
