#!/bin/bash
O=gpurun_out/r3b; mkdir -p $O
python tools/attn_variants_ab.py run nodeep deep8 ring4 ring4ps1 ring4ps2 --stages nodeep,ring4,ring4ps2 > $O/attn_ab.jsonl 2> $O/attn_ab.err; echo "ab rc=$?"
cat $O/attn_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    print({k: j[k] for k in j if k in ('lib','shape','us','GBps','outputs_differing_from_exact','mean_err_in_fp16_ulps','tag','retrieval_verify_us','target_verify_us','draft_step_us','ar_step_us','failed')})
"
python -m pytest tests/test_gpu_tp_offload.py tests/test_gpu_configs.py -q -k "bench_tp or full_size_cfg2 or error_path" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
