#!/bin/bash
# round 4, GPU call 4: GEMM + exchange in one launch (tf_skinny_gemm_xchg); norm-GEMM prologue load order A/B; new parity tests
set -x
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04c4
mkdir -p $O
R=$GRAFT_REPO_ROOT
L=$R/triforce_amd/lib
timeout 600 python -m pytest tests/test_gpu_tp_offload.py -x -q -k "gemm_exchange" 2>&1 | tail -15 > $O/pytest_xchg.txt
cat $O/pytest_xchg.txt
if grep -q "failed\|error" $O/pytest_xchg.txt; then echo "xchg tests failed: skipping engine runs that use it"; export TRIFORCE_TP_GEMM_XCHG=0; XFLAG=""; else XFLAG="--gemm-exchange"; fi
for cfg in "7B:llama-7B-128K 8 --gamma 6 --prefill 124928 --budget 4096" "13B:llama-13B-128K 8" "7Bw2:llama-7B-128K 2"; do
  tag=${cfg%%:*}; a=${cfg#*:}
  timeout 600 python tools/tp_shard_bench.py $a --local-exchange 2>$O/tp_${tag}_kernel.err | grep '^{' | sed "s/^{/{\"variant\": \"exchange kernel, round-4 prologue order\", /" >> $O/tp_shard.jsonl
  [ -n "$XFLAG" ] && timeout 600 python tools/tp_shard_bench.py $a --local-exchange $XFLAG 2>$O/tp_${tag}_xchg.err | grep '^{' | sed "s/^{/{\"variant\": \"GEMM+exchange in one launch\", /" >> $O/tp_shard.jsonl
  TRIFORCE_HIP_LIB=$L/libtriforce_hip_sgprol0.so timeout 600 python tools/tp_shard_bench.py $a --local-exchange 2>$O/tp_${tag}_prol0.err | grep '^{' | sed "s/^{/{\"variant\": \"exchange kernel, round-3 prologue order (SG_PROLOGUE_ORDER=0)\", /" >> $O/tp_shard.jsonl
done
timeout 600 python tools/verify_bench.py prologue_r4_7b_cfg2 2>$O/vb_r4.err | grep '^{' >> $O/verify_bench.jsonl
TRIFORCE_HIP_LIB=$L/libtriforce_hip_sgprol0.so timeout 600 python tools/verify_bench.py prologue_r3order_7b_cfg2 2>$O/vb_r3.err | grep '^{' >> $O/verify_bench.jsonl
timeout 600 python tools/verify_bench.py prologue_r4_7b_cfg2_again 2>>$O/vb_r4.err | grep '^{' >> $O/verify_bench.jsonl
if [ -n "$XFLAG" ]; then
(cd /tmp && export TMPDIR=/tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_tp8 -- python $R/tools/tp_shard_bench.py llama-7B-128K 8 --gamma 6 --prefill 124928 --budget 4096 --local-exchange --gemm-exchange > $R/$O/prof_tp8.log 2>&1)
T=$(ls -S $O/prof_tp8/*/*kernel_trace.csv | head -1)
python tools/kernel_timeline.py $T $O/tp8_7b_kernel_timeline.json "rocprofv3 --kernel-trace of tools/tp_shard_bench.py llama-7B-128K 8 --gamma 6 --prefill 124928 --budget 4096 --local-exchange --gemm-exchange (rank 0 shard of an 8-way 7B engine on one MI355X): o_proj / down_proj with the exchange in their epilogue, norm-GEMM prologue in round 4's load order" > $O/tp8_7b_kernel_timeline.txt 2>&1
find $O/prof_tp8 -name "*kernel_trace.csv" -size +20M -delete
fi
rm -f gpurun_out/parity_notes.txt
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > $O/pytest_gpu.txt
cp gpurun_out/parity_notes.txt $O/parity_notes.txt 2>/dev/null
cat $O/pytest_gpu.txt
for f in $O/*.err; do echo "== $f"; tail -n 2 $f; done
