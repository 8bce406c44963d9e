#!/bin/bash
O=gpurun_out/r2h; mkdir -p $O
python -m pytest tests/test_gpu_tp_offload.py tests/test_gpu_e2e.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log; grep "\[parity\]" $O/pytest.log
for nw in 0 32 64; do
  TF_ATTN_NW8=$nw timeout 200 python tools/verify_bench.py nw8_$nw 2>$O/vb_nw$nw.err | tee -a $O/verify_bench.jsonl
  TF_ATTN_NW8=$nw timeout 200 python tools/tune.py nw8_$nw 2>$O/tune_nw$nw.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:v for k,v in d.items() if 'retrieval' in k or k=='tag'})" | tee -a $O/tune_attn.txt
done
TF_ATTN_NW8=32 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "attn" > $O/pytest_nw8.log 2>&1; echo "pytest nw8 rc=$?"; tail -3 $O/pytest_nw8.log
