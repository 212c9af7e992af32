#!/bin/bash
# non-temporal default (mask 15) + non-temporal matrix stream for large matrices: kernel tests, bench, north-star A/B
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp; L=$O/r03_nt_ab2.log; : > $L
timeout 600 python -X faulthandler -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "csr or ritz or project or dots" 2>&1 | tail -2 >> $L
for env in "" "HIPK_SPMV_NT=1"; do
  echo "== bench [$env]" >> $L
  env $env timeout 300 python bench.py --no-cpu-baseline --no-north-star 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], {k:(v['ms'], v['GBps']) for k,v in r['all_kernels'].items()})" >> $L
done
for env in "HIPK_SPMV_NT=0" "" "HIPK_SPMV_NT=0" ""; do
  echo "== north star 3000 its [$env]" >> $L
  env $env timeout 300 python scripts/one_solve.py csr lap2d_10m 3000 2>&1 | tail -1 >> $L
done
cat $L
