#!/bin/bash
# Round 6, step 7: SpMM window (odd stride + the 1728-value buffer) against the round-5 layout on configs[2]; kernel tests.
R=$PWD; O=$R/gpurun_out; TAG=r06
python -m pytest tests/test_kernels_gpu.py tests/test_solver_gpu.py -q -m gpu -x -k "csr or jdqmr or extremal or subspace" 2>&1 | tail -3 > $O/${TAG}_s7_tests.log; cat $O/${TAG}_s7_tests.log
( for i in 1 2 3; do for v in odd even; do if [ $v = even ]; then export HIPK_SPMM_EVEN_STRIDE=1; else unset HIPK_SPMM_EVEN_STRIDE; fi
  echo "window stride $v"; python scripts/config3_run.py --reps 3 2>&1 | tail -1 | cut -c1-140; done; done ) > $O/${TAG}_spmm_window_stride2.txt 2>&1
unset HIPK_SPMM_EVEN_STRIDE; cat $O/${TAG}_spmm_window_stride2.txt
python scripts/spmm_perf.py 2>&1 | tail -12 > $O/${TAG}_spmm_perf.txt; cat $O/${TAG}_spmm_perf.txt
HIPK_SPMM_EVEN_STRIDE=1 python scripts/spmm_perf.py 2>&1 | tail -12 > $O/${TAG}_spmm_perf_even.txt; cat $O/${TAG}_spmm_perf_even.txt
