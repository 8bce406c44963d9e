#!/bin/bash
# round 4, GPU call 10: rendezvous merge of > 8 attention splits inside the launch (small-H tensor-parallel shards) —
# bit-identity tests, then TP-shard stage latencies with the rendezvous on / off
set -x
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04c10
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "attn" 2>&1 | tail -8 > $O/pytest_attn.txt
cat $O/pytest_attn.txt
for cfg in "7Bw8:llama-7B-128K 8 --gamma 6 --prefill 124928 --budget 4096" "7Bw4:llama-7B-128K 4 --gamma 6 --prefill 124928 --budget 4096" "13Bw8:llama-13B-128K 8"; do
  tag=${cfg%%:*}; a=${cfg#*:}
  for rep in 1 2; do
  timeout 600 python tools/tp_shard_bench.py $a --local-exchange --gemm-exchange 2>$O/tp_${tag}.err | grep '^{' | sed "s/^{/{\"variant\": \"rendezvous merge (default)\", /" >> $O/tp_shard.jsonl
  TRIFORCE_ATTN_RENDEZVOUS=0 timeout 600 python tools/tp_shard_bench.py $a --local-exchange --gemm-exchange 2>$O/tp_${tag}_off.err | grep '^{' | sed "s/^{/{\"variant\": \"TRIFORCE_ATTN_RENDEZVOUS=0 (merge kernel)\", /" >> $O/tp_shard.jsonl
  done
done
cat $O/tp_shard.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d.get('variant'), {k: v for k, v in d.items() if 'us' in k or k in ('model', 'world')})
"
for f in $O/*.err; do echo "== $f"; tail -n 2 $f; done
