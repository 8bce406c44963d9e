python tools/_probe_n8.py 2>&1 | tail -20
rm -f gpurun_out/parity_notes.txt
python -m pytest tests -m gpu -q -x > gpurun_out/pytest_full.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_full.log
