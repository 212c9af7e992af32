#!/bin/bash
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp; L=$O/r03_spmm_quick.log; : > $L
timeout 600 python -X faulthandler -m pytest tests/test_kernels_gpu.py -k "csr or slab or halo" -q -p no:cacheprovider 2>&1 | tail -2 >> $L
for env in "" "HIPK_SPMM_NO_COL16=1"; do
  echo "== spmm_perf [$env]" >> $L
  env $env timeout 300 python scripts/spmm_perf.py 2>&1 | grep '"matrix"' | cut -c1-200 >> $L
done
cat $L
