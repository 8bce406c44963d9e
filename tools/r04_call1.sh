#!/bin/bash
# round 4, GPU call 1: kernel rewrite sanity + the three sweeps of verdict item 1
set -x
mkdir -p gpurun_out/r04c1
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "skinny or qkv or swiglu or retrieval_score" 2>&1 | tail -15 > gpurun_out/r04c1/pytest_gemm.txt
timeout 600 python tools/gemm_layout_ab.py > gpurun_out/r04c1/gemm_layout_ab.jsonl 2> gpurun_out/r04c1/gemm_layout_ab.err
timeout 400 python tools/attn_nsplit_g16.py > gpurun_out/r04c1/attn_nsplit_g16.jsonl 2> gpurun_out/r04c1/attn_nsplit_g16.err
TRIFORCE_HIP_LIB=$GRAFT_REPO_ROOT/triforce_amd/lib/libtriforce_hip_rscoreceil.so timeout 200 python tools/attn_nsplit_g16.py score > gpurun_out/r04c1/score_oldrule.jsonl 2> gpurun_out/r04c1/score_oldrule.err
tail -5 gpurun_out/r04c1/*.err
cat gpurun_out/r04c1/pytest_gemm.txt
