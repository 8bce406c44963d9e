#!/bin/bash
O=gpurun_out/r2l; mkdir -p $O
python -m pytest tests/test_gpu_tp_offload.py tests/test_gpu_sequoia.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
for a in "llama-7B-128K 1 --gamma 6 --prefill 124928 --budget 4096" "llama-7B-128K 8 --gamma 6 --prefill 124928 --budget 4096" "llama-7B-128K 2" "llama-13B-128K 8"; do python tools/tp_shard_bench.py $a 2>>$O/shard.err | tee -a $O/shard.jsonl | cut -c1-330; done
