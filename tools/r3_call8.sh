#!/bin/bash
O=gpurun_out/r3g; mkdir -p $O
python tools/prefill_variants_ab.py default ahead2 ahead3 > $O/prefill_ab.jsonl 2> $O/prefill_ab.err; cat $O/prefill_ab.jsonl; tail -3 $O/prefill_ab.err
TRIFORCE_HIP_LIB=$PWD/triforce_amd/lib/libtriforce_hip_ahead3.so python -m pytest tests/test_gpu_ops.py -q -k "prefill or block" > $O/pytest_ahead3.log 2>&1; echo "pytest ahead3 rc=$?"; tail -3 $O/pytest_ahead3.log
