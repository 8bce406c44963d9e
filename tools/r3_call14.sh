#!/bin/bash
O=gpurun_out/r3j; mkdir -p $O
python tools/attn_merge_ab.py --fused-only "default (4-wave q2)" > $O/attn_default.jsonl 2> $O/err1
TRIFORCE_HIP_LIB=$PWD/triforce_amd/lib/libtriforce_hip_q2w8.so python tools/attn_merge_ab.py --fused-only "q2w8 (8-wave two-q-tile kernel)" > $O/attn_q2w8.jsonl 2> $O/err2
cat $O/attn_default.jsonl $O/attn_q2w8.jsonl | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    if j['sq'] > 16: print({k: j[k] for k in j if k in ('lib','shape','us','GBps','outputs_differing_from_exact','mean_err_in_fp16_ulps')})
"
TRIFORCE_HIP_LIB=$PWD/triforce_amd/lib/libtriforce_hip_q2w8.so python -m pytest tests/test_gpu_ops.py tests/test_gpu_configs.py -q -k "attn_decode" > $O/pytest_q2w8.log 2>&1; echo "pytest q2w8 rc=$?"; tail -3 $O/pytest_q2w8.log
python -m pytest tests/test_gpu_ops.py -q -k "attn_decode" > $O/pytest_default.log 2>&1; echo "pytest default rc=$?"; tail -2 $O/pytest_default.log
