#!/bin/bash
# A/B of the non-temporal load / store variants (build-time, scripts/build_variant.sh) on one box: bench line without the
# north-star solve and the CPU baseline, then the first 3000 outer iterations of the north-star workload
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp; L=$O/r03_nt_ab.log; : > $L
for v in "" nt3 nt7 nt15 nt31 ""; do
  if [ -z "$v" ]; then lib=""; else lib="PRIMME_AMD_LIB=$PWD/primme_amd/variants/libprimme_amd_$v.so"; fi
  echo "== bench [${v:-default}]" >> $L
  env $lib timeout 300 python bench.py --no-cpu-baseline --no-north-star 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], {k:(v['ms'], v['GBps']) for k,v in r['all_kernels'].items()})" >> $L
done
for v in "" nt3 nt7 nt15 nt31; do
  if [ -z "$v" ]; then lib=""; else lib="PRIMME_AMD_LIB=$PWD/primme_amd/variants/libprimme_amd_$v.so"; fi
  echo "== north star 3000 its [${v:-default}]" >> $L
  env $lib timeout 300 python scripts/one_solve.py csr lap2d_10m 3000 2>&1 | tail -1 >> $L
done
cat $L
