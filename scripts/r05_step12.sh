#!/bin/bash
# round 5, step 12: the full GPU suite with the normal interpreter exit (run 4 of the round), smoke()
cd "$(dirname "$0")/.."
O=gpurun_out/r05_step12; mkdir -p $O
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider > $O/full_suite.txt 2>&1; echo "full suite (normal interpreter exit) exit code $?" | tee -a $O/full_suite.txt
tail -6 $O/full_suite.txt
cat gpurun_out/exact_history_gpu.json; echo
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3
