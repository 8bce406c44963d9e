#!/bin/bash
# round 4, GPU call 3: split-K GEMMs across workgroups + many-split in-launch attention merge at TP-shard shapes; P hi+lo
# in the block / prefill kernels; new TP tests (segments + alternating halves, litmus dry run)
set -x
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04c3
mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -k "split_across or one_launch_merge or shard_shapes or fused_merge_never" 2>&1 | tail -15 > $O/pytest_new_kernels.txt
timeout 1200 python -m pytest tests/test_gpu_tp_offload.py -x -q -k "segment_graphs or litmus or rccl_world2 or world2_on_one_device" 2>&1 | tail -15 > $O/pytest_tp_new.txt
L=triforce_amd/lib
for cfg in "7B:llama-7B-128K 8 --gamma 6 --prefill 124928 --budget 4096" "13B:llama-13B-128K 8" "7Bw2:llama-7B-128K 2" "7Bw4:llama-7B-128K 4 --gamma 6 --prefill 124928 --budget 4096"; do
  tag=${cfg%%:*}; a=${cfg#*:}
  timeout 600 python tools/tp_shard_bench.py $a --local-exchange 2>$O/tp_${tag}_new.err | grep '^{' | sed "s/^{/{\"variant\": \"split-K GEMMs + in-launch many-split merge\", /" >> $O/tp_shard.jsonl
  TRIFORCE_GEMM_KSPLIT=0 timeout 600 python tools/tp_shard_bench.py $a --local-exchange 2>$O/tp_${tag}_noks.err | grep '^{' | sed "s/^{/{\"variant\": \"TRIFORCE_GEMM_KSPLIT=0\", /" >> $O/tp_shard.jsonl
  TRIFORCE_HIP_LIB=$R/$L/libtriforce_hip_nobigmerge.so timeout 600 python tools/tp_shard_bench.py $a --local-exchange 2>$O/tp_${tag}_nomerge.err | grep '^{' | sed "s/^{/{\"variant\": \"two-launch merge above 8 splits\", /" >> $O/tp_shard.jsonl
done
(cd /tmp && export TMPDIR=/tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_tp8 -- python $R/tools/tp_shard_bench.py llama-7B-128K 8 --gamma 6 --prefill 124928 --budget 4096 --local-exchange > $R/$O/prof_tp8.log 2>&1)
T=$(ls -S $O/prof_tp8/*/*kernel_trace.csv | head -1)
python tools/kernel_timeline.py $T $O/tp8_7b_kernel_timeline.json "rocprofv3 --kernel-trace of tools/tp_shard_bench.py llama-7B-128K 8 --gamma 6 --prefill 124928 --budget 4096 --local-exchange (rank 0 shard of an 8-way 7B engine on one MI355X), round-4 build: K split across workgroups in the q|k|v / gate|up GEMMs, many-split attention merge inside the launch" > $O/tp8_7b_kernel_timeline.txt 2>&1
find $O/prof_tp8 -name "*kernel_trace.csv" -size +20M -delete
timeout 900 python tools/prefill_variants_ab.py default psplitblk default psplitblk > $O/psplit_block_ab.jsonl 2> $O/psplit_block_ab.err
cat $O/pytest_new_kernels.txt $O/pytest_tp_new.txt
for f in $O/*.err; do echo "== $f"; tail -n 2 $f; done
