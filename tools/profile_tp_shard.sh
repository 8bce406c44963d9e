#!/bin/bash
# rocprofv3 kernel trace of rank 0's shard of a W-way engine on one GPU -> per-kernel timeline of a decode layer
#   bash tools/profile_tp_shard.sh <label> [env VAR=VALUE ...] -- <tp_shard_bench args>     -> gpurun_out/tpprof/<label>_timeline.json
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/tpprof; mkdir -p $O
label=$1; shift; envs=()
while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
(cd /tmp && export TMPDIR=/tmp && env "${envs[@]}" timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$label -- python $R/tools/tp_shard_bench.py "$@" > $O/$label.log 2>&1)
T=$(ls -S $O/$label/*/*kernel_trace.csv | head -1)
python $R/tools/kernel_timeline.py $T $O/${label}_timeline.json "rocprofv3 --kernel-trace of tools/tp_shard_bench.py $* (${envs[*]})" > $O/${label}_timeline.txt 2>&1
grep '^{' $O/$label.log | tail -1 > $O/${label}_bench.json
find $O/$label -name "*kernel_trace.csv" -size +20M -delete
head -40 $O/${label}_timeline.txt
