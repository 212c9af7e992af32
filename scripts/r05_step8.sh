#!/bin/bash
# round 5, step 8: row-pattern SpMV with the branch-free trip (end-of-slab tests only in the last chunk), A/B over rows per lane / resident waves
cd "$(dirname "$0")/.."
O=gpurun_out/r05_step8; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -p no:cacheprovider -k "pattern or csr or halo or matvec" > $O/kernel_tests.txt 2>&1; echo "kernel tests exit $?"; tail -3 $O/kernel_tests.txt
for v in default r1_w8 r1_w5 r1_w4 r2_w4; do
  if [ $v = default ]; then unset PRIMME_AMD_LIB; else export PRIMME_AMD_LIB=$PWD/primme_amd/variants/libprimme_amd_pat_$v.so; fi
  echo "== $v"; timeout 300 python scripts/spmv_format_perf.py 100 2>&1 | grep -v amdgpu.ids | grep "format 2\|bit\|differ" | tee -a $O/perf_$v.txt
done
