#!/bin/bash
# round 5, step 14: the tree after the final evidence pass (QMR step statistics): configs[2] full-size test, QMR kernels, JDQMR fixtures
cd "$(dirname "$0")/.."
O=gpurun_out/r05_step14; mkdir -p $O
timeout 900 python -m pytest tests/test_full_size_configs_gpu.py tests/test_kernels_gpu.py tests/test_solver_gpu.py -m gpu -q -x -p no:cacheprovider -k "config3 or qmr or jdqmr or blk" > $O/tests.txt 2>&1; echo "tests exit $?"; tail -3 $O/tests.txt
