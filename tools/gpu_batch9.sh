#!/bin/bash
O=gpurun_out/r2i; mkdir -p $O
for v in default u8 w16 u2 w16u8 default; do
  if [ $v = default ]; then unset TRIFORCE_HIP_LIB; else export TRIFORCE_HIP_LIB=$PWD/triforce_amd/lib/libtriforce_hip_$v.so; fi
  timeout 200 python tools/tune.py $v gemm 2>$O/tune_$v.err | tee -a $O/tune_gemm.jsonl | cut -c1-700
  timeout 200 python tools/verify_bench.py $v 2>$O/vb_$v.err | tee -a $O/verify_bench.jsonl
done
