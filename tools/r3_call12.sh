#!/bin/bash
O=gpurun_out/r3i; mkdir -p $O
for C in 1024 2048 4096; do TRIFORCE_PREFILL_CHUNK=$C python tools/prefill_time.py 2>/dev/null | grep "^{" ; done | tee $O/prefill_chunk.jsonl
