#!/bin/bash
# round 5, step 13: block QMR step with one host synchronisation (scalar recurrences on the device): kernels, solver tests, configs[2] A/B
cd "$(dirname "$0")/.."
O=gpurun_out/r05_step13; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -p no:cacheprovider -k "qmr" > $O/kernel_tests.txt 2>&1; echo "kernel tests exit $?"; tail -2 $O/kernel_tests.txt
timeout 900 python -m pytest tests/test_solver_gpu.py tests/test_full_size_configs_gpu.py -m gpu -q -x -p no:cacheprovider -k "jdqmr or JDQMR or blk or config3 or configs2 or lunda" > $O/solver_tests.txt 2>&1; echo "solver tests exit $?"; tail -2 $O/solver_tests.txt
for mode in one three one three; do
  if [ $mode = three ]; then export PRIMME_AMD_QMR_THREE_WAITS=1; else unset PRIMME_AMD_QMR_THREE_WAITS; fi
  echo "== $mode"; timeout 300 python scripts/config3_run.py --reps 3 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-330 | tee -a $O/configs2_$mode.txt
done
