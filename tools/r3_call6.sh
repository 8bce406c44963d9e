#!/bin/bash
O=gpurun_out/r3f; mkdir -p $O
python tools/prefill_variants_ab.py default pipe1 pipe2 pipe3 > $O/prefill_ab.jsonl 2> $O/prefill_ab.err; cat $O/prefill_ab.jsonl
python -m pytest tests/test_gpu_ops.py -q -k "prefill or block or tree" > $O/pytest_default.log 2>&1; echo "pytest default rc=$?"; tail -3 $O/pytest_default.log
TRIFORCE_HIP_LIB=$PWD/triforce_amd/lib/libtriforce_hip_pipe2.so python -m pytest tests/test_gpu_ops.py -q -k "prefill or block" > $O/pytest_pipe2.log 2>&1; echo "pytest pipe2 rc=$?"; tail -3 $O/pytest_pipe2.log
