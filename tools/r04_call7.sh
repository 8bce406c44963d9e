#!/bin/bash
# round 4, GPU call 7: GEMM + exchange with per-panel epochs / no acquire fence; phase stamps of the TP-shard norm GEMMs;
# the tests added since the last full run
set -x
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04c7
mkdir -p $O
rm -f gpurun_out/parity_notes.txt
timeout 1500 python -m pytest tests/test_gpu_tp_offload.py tests/test_gpu_ops.py tests/test_gpu_e2e.py -q -k "gemm_exchange or world2_on_one_device or segment_graphs or down_proj or norm_prologue or expected_acceptance or split_across or bench_tp_line" 2>&1 | tail -12 > $O/pytest_sel.txt
cat $O/pytest_sel.txt
cp gpurun_out/parity_notes.txt $O/parity_notes.txt 2>/dev/null
for cfg in "7Bw8:llama-7B-128K 8 --gamma 6 --prefill 124928 --budget 4096" "13Bw8:llama-13B-128K 8" "7Bw2g16:llama-7B-128K 2"; do
  tag=${cfg%%:*}; a=${cfg#*:}
  timeout 600 python tools/tp_shard_bench.py $a --local-exchange --gemm-exchange 2>$O/tp_${tag}.err | grep '^{' | sed "s/^{/{\"variant\": \"GEMM+exchange: per-panel epochs, no acquire fence\", /" >> $O/tp_shard.jsonl
  timeout 600 python tools/tp_shard_bench.py $a --local-exchange 2>$O/tp_${tag}_k.err | grep '^{' | sed "s/^{/{\"variant\": \"exchange kernel\", /" >> $O/tp_shard.jsonl
done
timeout 300 python tools/gemm_stamps.py > $O/gemm_phase_stamps_7rows.json 2> $O/stamps7.err
STAMP_ROWS=17 timeout 300 python tools/gemm_stamps.py > $O/gemm_phase_stamps_17rows.json 2> $O/stamps17.err
cat $O/gemm_phase_stamps_7rows.json
for f in $O/*.err; do echo "== $f"; tail -n 2 $f; done
