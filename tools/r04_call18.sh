#!/bin/bash
# round 4, GPU call 18: the 68M draft's lm_head (N 32000, K 768: 30.9 us for 49 MB) under the other launch forms —
# one panel per wave (2000 workgroups), two panels with 8 waves — via the draft step latency; + 13B cfg4 stage latencies
# to see what the same switch does to the shapes the two-panel rule was measured on
set -x
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04c18
mkdir -p $O
run() {
  tag=$1; shift
  env "$@" timeout 300 python tools/verify_bench.py "$tag" 2>$O/vb_$tag.err | grep '^{' | sed "s/^{/{\"env\": \"$*\", /" >> $O/verify_bench_p2.jsonl
}
run default A=1
run p2off TRIFORCE_GEMM_P2_GROUPS=1001
run p2w8 TRIFORCE_GEMM_P2_WAVES=8
run default2 A=2
run p2off2 TRIFORCE_GEMM_P2_GROUPS=1001
cat $O/verify_bench_p2.jsonl
tail -n 2 $O/*.err | cut -c1-200
