#!/bin/bash
O=gpurun_out/r2c; mkdir -p $O
for pf in 0 1 2 4 8; do
  TRIFORCE_PREFETCH_NEXT=$pf timeout 200 python tools/verify_bench.py pf$pf 2>$O/vb_pf$pf.err | tee -a $O/verify_bench.jsonl
done
# tree-mask funnel variant: Sequoia kernels A/B + its tests under the variant library
timeout 200 python tools/tune.py default > $O/tune_default.json 2> $O/tune_default.err; tail -c 1500 $O/tune_default.json
TRIFORCE_HIP_LIB=$PWD/triforce_amd/lib/libtriforce_hip_funnel.so timeout 200 python tools/tune.py funnel > $O/tune_funnel.json 2> $O/tune_funnel.err; tail -c 600 $O/tune_funnel.json
TRIFORCE_HIP_LIB=$PWD/triforce_amd/lib/libtriforce_hip_funnel.so timeout 300 python -m pytest tests/test_gpu_sequoia.py -m gpu -q > $O/pytest_funnel.log 2>&1; tail -3 $O/pytest_funnel.log
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
