#!/bin/bash
# round 4, GPU call 6: 16 waves per panel for few-panel GEMMs (TP shards), batched K-loop tail, 3-way split of the 13B
# down_proj in situ; GEMM / layout / TP tests on the new build
set -x
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04c6
mkdir -p $O
R=$GRAFT_REPO_ROOT
L=$R/triforce_amd/lib
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_layouts.py -x -q -k "gemm or skinny or swiglu or qkv or layouts or exchange or down_proj" 2>&1 | tail -12 > $O/pytest_gemm.txt
cat $O/pytest_gemm.txt
for cfg in "7Bw8:llama-7B-128K 8 --gamma 6 --prefill 124928 --budget 4096" "7Bw4:llama-7B-128K 4 --gamma 6 --prefill 124928 --budget 4096" "13Bw8:llama-13B-128K 8" "7Bw2g16:llama-7B-128K 2"; do
  tag=${cfg%%:*}; a=${cfg#*:}
  timeout 600 python tools/tp_shard_bench.py $a --local-exchange --gemm-exchange 2>$O/tp_${tag}_few.err | grep '^{' | sed "s/^{/{\"variant\": \"16 \/ 8 waves per panel up to 200 panels, batched tail\", /" >> $O/tp_shard.jsonl
  TRIFORCE_GEMM_FEW_PANELS=0 timeout 600 python tools/tp_shard_bench.py $a --local-exchange --gemm-exchange 2>$O/tp_${tag}_nofew.err | grep '^{' | sed "s/^{/{\"variant\": \"TRIFORCE_GEMM_FEW_PANELS=0 (round-3 wave counts), batched tail\", /" >> $O/tp_shard.jsonl
  TRIFORCE_HIP_LIB=$L/libtriforce_hip_sgtail0.so TRIFORCE_GEMM_FEW_PANELS=0 timeout 600 python tools/tp_shard_bench.py $a --local-exchange --gemm-exchange 2>$O/tp_${tag}_tail0.err | grep '^{' | sed "s/^{/{\"variant\": \"round-3 wave counts, chunk-by-chunk tail (SG_TAIL_BATCH=0)\", /" >> $O/tp_shard.jsonl
done
timeout 600 python tools/verify_bench.py tail_batch_7b_cfg2 2>$O/vb1.err | grep '^{' >> $O/verify_bench.jsonl
TRIFORCE_HIP_LIB=$L/libtriforce_hip_sgtail0.so timeout 600 python tools/verify_bench.py tail_scalar_7b_cfg2 2>$O/vb2.err | grep '^{' >> $O/verify_bench.jsonl
timeout 600 python tools/verify_bench.py tail_batch_7b_cfg2_again 2>>$O/vb1.err | grep '^{' >> $O/verify_bench.jsonl
timeout 900 python tools/verify_bench.py ks3_13b_cfg4 --target llama-13B-128K --prefill 130048 --budget 12288 --gamma 16 2>$O/vb3.err | grep '^{' >> $O/verify_bench.jsonl
(cd /tmp && export TMPDIR=/tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_tp8 -- python $R/tools/tp_shard_bench.py llama-7B-128K 8 --gamma 6 --prefill 124928 --budget 4096 --local-exchange --gemm-exchange > $R/$O/prof_tp8.log 2>&1)
T=$(ls -S $O/prof_tp8/*/*kernel_trace.csv | head -1)
python tools/kernel_timeline.py $T $O/tp8_7b_kernel_timeline.json "rocprofv3 --kernel-trace of tools/tp_shard_bench.py llama-7B-128K 8 --gamma 6 --prefill 124928 --budget 4096 --local-exchange --gemm-exchange (rank 0 shard of an 8-way 7B engine on one MI355X): GEMM + exchange in one launch, 16 waves per panel in the q|k|v / gate|up shards" > $O/tp8_7b_kernel_timeline.txt 2>&1
find $O/prof_tp8 -name "*kernel_trace.csv" -size +20M -delete
for f in $O/*.err; do echo "== $f"; tail -n 2 $f; done
