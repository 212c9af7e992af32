#!/bin/bash
# round 5, first measurement call: the row-pattern SpMV form (kernel tests, isolated timings, the solver fixtures on it) and bench.py with all configs
cd "$(dirname "$0")/.."
O=gpurun_out/r05_step1; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "csr or pattern" > $O/kernel_tests.txt 2>&1; echo "kernel tests exit $?" | tee -a $O/kernel_tests.txt
tail -5 $O/kernel_tests.txt
timeout 300 python scripts/spmv_format_perf.py > $O/spmv_format_perf.txt 2>&1; echo "perf exit $?"; cat $O/spmv_format_perf.txt
timeout 900 python -m pytest tests/test_solver_gpu.py tests/test_svds_gpu.py tests/test_reference_kernels.py -m gpu -q -p no:cacheprovider > $O/solver_tests.txt 2>&1; echo "solver tests exit $?" | tee -a $O/solver_tests.txt
tail -8 $O/solver_tests.txt
timeout 900 python bench.py --steps 1 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; tail -c 3000 $O/bench.err; cat $O/bench.json
