#!/bin/bash
# round 5, step 17: solver fixtures + smoke on the final library
cd "$(dirname "$0")/.."
O=gpurun_out/r05_step17; mkdir -p $O
timeout 200 python -m pytest tests/test_solver_gpu.py -m gpu -q -p no:cacheprovider > $O/solver.txt 2>&1; echo "solver tests exit $?"; tail -2 $O/solver.txt
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1
