#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3e; mkdir -p $O
python tools/prefill_variants_ab.py default pipe1 pipe2 > $O/prefill_ab.jsonl 2> $O/prefill_ab.err; cat $O/prefill_ab.jsonl
cd /tmp && export TMPDIR=/tmp
for L in default pipe2; do
  LIB=$R/triforce_amd/lib/libtriforce_hip.so; [ $L != default ] && LIB=$R/triforce_amd/lib/libtriforce_hip_$L.so
  TRIFORCE_HIP_LIB=$LIB rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $O/p1_$L -o p1 --output-format csv -- python $R/tools/prefill_attn_once.py > $O/p1_$L.log 2>&1; echo p1 $L rc=$?
  TRIFORCE_HIP_LIB=$LIB rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES --kernel-trace -d $O/p2_$L -o p2 --output-format csv -- python $R/tools/prefill_attn_once.py > $O/p2_$L.log 2>&1; echo p2 $L rc=$?
done
cd $R
python - <<'PY'
import csv, glob, json, os
from collections import defaultdict
O = "gpurun_out/r3e"
for L in ("default", "pipe2"):
    acc, n = defaultdict(float), defaultdict(int)
    for f in glob.glob(f"{O}/p?_{L}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn_prefill_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    disp = max(1, len({0}))
    print(L, json.dumps({k: round(v / 3, 0) for k, v in sorted(acc.items())}))
PY
find $O -name "*kernel_trace.csv" -delete
python tools/acceptance_sweep.py gpurun_out/r3e/acceptance_sweep.json --steps 24 > gpurun_out/r3e/acceptance_sweep.log 2>&1; echo "sweep rc=$?"; tail -9 gpurun_out/r3e/acceptance_sweep.log | cut -c1-330
