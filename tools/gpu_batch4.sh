#!/bin/bash
O=gpurun_out/r2d; mkdir -p $O
for v in default nopipe pipe_u8 pipe_w16; do
  if [ $v = default ]; then unset TRIFORCE_HIP_LIB; else export TRIFORCE_HIP_LIB=$PWD/triforce_amd/lib/libtriforce_hip_$v.so; fi
  timeout 200 python tools/tune.py $v gemm 2>$O/tune_$v.err | tee -a $O/tune_gemm.jsonl | cut -c1-900
  timeout 200 python tools/verify_bench.py $v 2>$O/vb_$v.err | tee -a $O/verify_bench.jsonl
done
export TRIFORCE_HIP_LIB=$PWD/triforce_amd/lib/libtriforce_hip_funnel.so
timeout 200 python tools/tune.py funnel 2>$O/tune_funnel.err | tee $O/tune_funnel.json | cut -c1-1600
timeout 300 python -m pytest tests/test_gpu_sequoia.py -m gpu -q > $O/pytest_funnel.log 2>&1; tail -3 $O/pytest_funnel.log
unset TRIFORCE_HIP_LIB
timeout 200 python tools/tune.py default2 2>>$O/tune_default.err | tee $O/tune_default_full.json | cut -c1-1600
python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
