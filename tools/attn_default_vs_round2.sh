#!/bin/bash
O=gpurun_out/attn_vs_r2; mkdir -p $O
rm -f gpurun_out/parity_notes.txt
python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
cp gpurun_out/parity_notes.txt $O/parity_notes.txt
python tools/attn_merge_ab.py --fused-only "default (P hi+lo)" > $O/attn_default.jsonl 2> $O/attn_default.err
TRIFORCE_HIP_LIB=triforce_amd/lib/libtriforce_hip_ps0.so python tools/attn_merge_ab.py --fused-only "ps0 (round 2 kernel)" > $O/attn_ps0.jsonl 2> $O/attn_ps0.err
cat $O/attn_default.jsonl $O/attn_ps0.jsonl | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    print({k: j[k] for k in j if k in ('lib','shape','us','GBps','outputs_differing_from_exact','mean_err_in_fp16_ulps')})
"
python tools/verify_bench.py default > $O/verify_default.json 2>/dev/null; cat $O/verify_default.json
TRIFORCE_HIP_LIB=triforce_amd/lib/libtriforce_hip_ps0.so python tools/verify_bench.py ps0 > $O/verify_ps0.json 2>/dev/null; cat $O/verify_ps0.json
cat $O/parity_notes.txt
