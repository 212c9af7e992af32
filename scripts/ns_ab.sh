#!/bin/bash
# north-star workload, first 3000 outer iterations: A/B of the round-3 knobs on one box
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp; L=$O/r03_ns_ab.log; : > $L
for env in "" "HIPK_SPMM_NO_COL16=1" "HIPK_NO_COPY_KERNEL=1" ""; do
  echo "== [$env]" >> $L
  env $env timeout 600 python scripts/one_solve.py csr lap2d_10m 3000 2>&1 | tail -1 >> $L
done
cat $L
