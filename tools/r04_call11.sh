#!/bin/bash
# round 4, GPU call 11: fixed step costs — token ids as kernel arguments (tf_set_tokens), K + V row copies / shifts in one
# launch, catch-up draft forward issued first: decode tests, bench line, idle-gap analysis of the same command under rocprofv3
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R"
O=gpurun_out/r04c11
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest.txt
cat $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-steps 0 > $O/bench2.json 2> $O/bench2.err; echo "bench2 rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-steps 0 > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err; echo "rocprof rc=$?"
T=$(ls -S $R/$O/prof/*/*kernel_trace.csv | head -1)
python $R/tools/gap_analysis.py $T --steps 19 > $R/$O/gap_analysis_decode_steps.txt 2>&1; head -40 $R/$O/gap_analysis_decode_steps.txt | cut -c1-160
find $R/$O/prof -name "*kernel_trace.csv" -size +20M -delete
cd $R
python bench.py --target llama-13B-128K --prefill 130048 --budget 12288 --gamma 16 --steps 10 --warmup 2 --no-cpu-baseline --random-steps 0 > $O/bench_13b_cfg4.json 2> $O/bench_13b.err; echo "13B rc=$?"
python - <<'PY'
import json
for f in ("bench", "bench2", "bench_prof", "bench_13b_cfg4"):
    try:
        d = json.load(open(f"gpurun_out/r04c11/{f}.json"))
        print(f, d["value"], d["ms_per_step"], d["step_overhead_us"], d["stage_latency_us"])
    except Exception as e:
        print(f, "failed", e)
PY
tail -3 $O/*.err
