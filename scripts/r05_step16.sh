#!/bin/bash
# round 5, step 16: row-pattern kernel — pairs whose rows differ in pattern reload both rows entry by entry; matrices with full chunks only
cd "$(dirname "$0")/.."
O=gpurun_out/r05_step16; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "pattern or csr_matvec or halo" > $O/kernel_tests.txt 2>&1; echo "kernel tests exit $?"; tail -3 $O/kernel_tests.txt
timeout 120 python scripts/spmv_format_perf.py 60 2>&1 | grep -v amdgpu.ids | grep "format 2\|bit" | tee $O/perf.txt
