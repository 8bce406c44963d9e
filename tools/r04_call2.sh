#!/bin/bash
# round 4, GPU call 2: k-octet-major activations — parity tests, in-situ stage latencies by layout, TP-shard forms, full suite
set -x
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04c2
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_layouts.py -x -q 2>&1 | tail -25 > $O/pytest_layouts.txt
for lay in rows packed; do
  TRIFORCE_ACT_LAYOUT=$lay timeout 600 python tools/verify_bench.py ${lay}_7b_cfg2 2>$O/vb_${lay}_7b.err | grep '^{' >> $O/verify_bench.jsonl
  TRIFORCE_ACT_LAYOUT=$lay timeout 600 python tools/verify_bench.py ${lay}_7b_g16 --prefill 130048 --budget 12288 --gamma 16 2>$O/vb_${lay}_7bg16.err | grep '^{' >> $O/verify_bench.jsonl
  TRIFORCE_ACT_LAYOUT=$lay timeout 900 python tools/verify_bench.py ${lay}_13b_cfg4 --target llama-13B-128K --prefill 130048 --budget 12288 --gamma 16 2>$O/vb_${lay}_13b.err | grep '^{' >> $O/verify_bench.jsonl
done
for lay in rows packed; do
  TRIFORCE_ACT_LAYOUT=$lay timeout 600 python tools/tp_shard_bench.py llama-7B-128K 8 --gamma 6 --prefill 124928 --budget 4096 --local-exchange 2>$O/tp_${lay}_7b8.err | grep '^{' | sed "s/^{/{\"layout\": \"$lay\", /" >> $O/tp_shard.jsonl
  TRIFORCE_ACT_LAYOUT=$lay timeout 600 python tools/tp_shard_bench.py llama-13B-128K 8 --local-exchange 2>$O/tp_${lay}_13b8.err | grep '^{' | sed "s/^{/{\"layout\": \"$lay\", /" >> $O/tp_shard.jsonl
  TRIFORCE_ACT_LAYOUT=$lay timeout 600 python tools/tp_shard_bench.py llama-7B-128K 2 --local-exchange 2>$O/tp_${lay}_7b2.err | grep '^{' | sed "s/^{/{\"layout\": \"$lay\", /" >> $O/tp_shard.jsonl
done
TRIFORCE_TP_FUSE_MAX_ROWS=16 timeout 600 python tools/tp_shard_bench.py llama-13B-128K 8 --local-exchange 2>$O/tp_packed16_13b8.err | grep '^{' | sed "s/^{/{\"layout\": \"packed, fused <= 16 rows\", /" >> $O/tp_shard.jsonl
R=$GRAFT_REPO_ROOT
(cd /tmp && export TMPDIR=/tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_tp8 -- python $R/tools/tp_shard_bench.py llama-7B-128K 8 --gamma 6 --prefill 124928 --budget 4096 --local-exchange > $R/$O/prof_tp8.log 2>&1)
T=$(ls -S $O/prof_tp8/*/*kernel_trace.csv | head -1)
python tools/kernel_timeline.py $T $O/tp8_7b_kernel_timeline.json "rocprofv3 --kernel-trace of tools/tp_shard_bench.py llama-7B-128K 8 --gamma 6 --prefill 124928 --budget 4096 --local-exchange (rank 0 shard of an 8-way 7B engine on one MI355X)" > $O/tp8_7b_kernel_timeline.txt 2>&1
find $O/prof_tp8 -name "*kernel_trace.csv" -size +20M -delete
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > $O/pytest_gpu.txt
cat $O/pytest_layouts.txt $O/pytest_gpu.txt
for f in $O/*.err; do echo "== $f"; tail -n 3 $f; done
