# SQ counters of the prefill attention kernel (two rocprofv3 --pmc passes, kernel-trace only).
#   gpurun --timeout 600 -- "bash tools/prefill_pmc.sh"   -> gpurun_out/r2k/p{1,2}/*_counter_collection.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2k; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $O/p1 -o p1 --output-format csv -- python $R/tools/prefill_attn_once.py > $O/p1.log 2>&1; echo p1 rc=$?
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES --kernel-trace -d $O/p2 -o p2 --output-format csv -- python $R/tools/prefill_attn_once.py > $O/p2.log 2>&1; echo p2 rc=$?
find $O -name "*counter_collection.csv" | head; tail -3 $O/p1.log
