#!/bin/bash
O=gpurun_out/r2g; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log; grep "\[parity\]" $O/pytest.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -- python $R/tools/pmc_attn.py > $R/$O/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -- python $R/tools/pmc_attn.py > $R/$O/pmc_write.log 2>&1; echo "pmc write rc=$?"
cd $R; python tools/pmc_reduce.py $O/pmc_fetch $O/pmc_write $O/pmc_attn_target_verify.json "round 2"
find $O/pmc_fetch $O/pmc_write -name "*.csv" | head; find $O -name "*kernel_trace.csv" -size +20M -delete
