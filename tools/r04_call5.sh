#!/bin/bash
# round 4, GPU call 5: full GPU suite on the current build; per-rank stage latencies at W = 1/2/4/8 for configs[1]/[3]/[4]
# (inputs of tools/predict_scaling.py); forced split-K on the 320-panel 13B GEMMs; attention split counts at 20 / 10 heads
set -x
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04c5
mkdir -p $O
rm -f gpurun_out/parity_notes.txt
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -30 > $O/pytest_gpu.txt
cp gpurun_out/parity_notes.txt $O/parity_notes.txt 2>/dev/null
cat $O/pytest_gpu.txt
for W in 1 2 4 8; do
  X="--local-exchange --gemm-exchange"; [ $W = 1 ] && X=""
  timeout 500 python tools/tp_shard_bench.py llama-7B-128K $W --gamma 6 --prefill 124928 --budget 4096 $X 2>$O/sc_cfg1_w$W.err | grep '^{' >> $O/tp_shard_by_world.jsonl
  timeout 500 python tools/tp_shard_bench.py llama-7B-128K $W --gamma 16 --prefill 130048 --budget 12288 $X 2>$O/sc_cfg3_w$W.err | grep '^{' >> $O/tp_shard_by_world.jsonl
  timeout 700 python tools/tp_shard_bench.py llama-13B-128K $W --gamma 16 --prefill 130048 --budget 12288 $X 2>$O/sc_cfg4_w$W.err | grep '^{' >> $O/tp_shard_by_world.jsonl
done
GEMM_ONLY="13B:o,13B:down,7B:o,7B:down" GEMM_RULES="p1,p1ks2,p1ks3" GEMM_ROWS="8,17,32" timeout 300 python tools/gemm_layout_ab.py > $O/gemm_ksplit_force_ab.jsonl 2> $O/gemm_ksplit_force_ab.err
for f in $O/*.err; do echo "== $f"; tail -n 2 $f; done
