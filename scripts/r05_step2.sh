#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_step2; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "csr or pattern" > $O/kernel_tests.txt 2>&1; echo "kernel tests exit $?"; tail -3 $O/kernel_tests.txt
for v in default pat_r2_w8 pat_r2_w6 pat_r4_w4 pat_r8_w4; do
  if [ $v = default ]; then unset PRIMME_AMD_LIB; else export PRIMME_AMD_LIB=$PWD/primme_amd/variants/libprimme_amd_$v.so; fi
  echo "=== $v" | tee -a $O/spmv_format_perf.txt
  timeout 300 python scripts/spmv_format_perf.py 2>&1 | grep -v amdgpu.ids | tee -a $O/spmv_format_perf.txt | grep -E "format 2|identical"
done
unset PRIMME_AMD_LIB
timeout 900 python bench.py --steps 1 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; grep -v amdgpu.ids $O/bench.err | tail -c 2000; cat $O/bench.json
