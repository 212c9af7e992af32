#!/bin/bash
# Round 3: 2-byte index stream in the 1-column stream kernel and the window SpMM (template parameter), small window buffer,
# two lanes per row: tests, A/B timing (SpMM, configs[2]), bench line
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
L=$O/r03_spmm_campaign.log
: > $L
timeout 600 python -X faulthandler -m pytest tests/test_kernels_gpu.py -k "csr or stencil or slab or halo" -q -p no:cacheprovider > $O/r03_spmm_kernels.log 2>&1; echo "rc=$?" >> $O/r03_spmm_kernels.log
tail -3 $O/r03_spmm_kernels.log >> $L
for env in "" "HIPK_SPMM_NO_COL16=1"; do
  echo "== spmm_perf [$env]" >> $L
  env $env timeout 300 python scripts/spmm_perf.py 2>&1 | grep '"matrix"' | cut -c1-200 >> $L
done
echo "== configs[2]" >> $L
timeout 300 python scripts/config3_run.py 2>&1 | tail -1 | cut -c1-120 >> $L
HIPK_SPMM_NO_COL16=1 timeout 300 python scripts/config3_run.py 2>&1 | tail -1 | cut -c1-120 >> $L
echo "== bench" >> $L
timeout 600 python bench.py 2>&1 | tail -1 > $O/r03_bench_line_col16.json
python -c "
import json,sys; d=json.load(open('$O/r03_bench_line_col16.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['all_kernels']['csr_stream_kernel'], d['north_star']['seconds_per_solve'], d['north_star']['roofline']['all_kernels']['csr_stream_kernel'])" >> $L
cat $L
