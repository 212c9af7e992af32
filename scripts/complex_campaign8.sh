#!/bin/bash
# Round 3, eighth GPU pass: LDS row padding in the multi-lane restart kernel
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
L=$O/r03_complex_campaign8.log
: > $L
timeout 900 python -X faulthandler -m pytest tests/test_kernels_complex_gpu.py -k "ritz" -q -p no:cacheprovider -x > $O/r03_c8_kernels.log 2>&1; echo "rc=$?" >> $O/r03_c8_kernels.log
tail -4 $O/r03_c8_kernels.log >> $L
echo "== configs[3] native" >> $L
FORM=native timeout 300 python scripts/config4_run.py 2>&1 | cut -c1-120 >> $L
FORM=native timeout 300 rocprofv3 --kernel-trace -d $O/r03_prof_c8 -o c8 -- python scripts/config4_run.py > $O/r03_config4_under_rocprof.log 2>&1
python scripts/rocpd_gaps.py $O/r03_prof_c8/c8_results.db $O/r03_config4_native_gaps.md 2>&1 | head -3 >> $L
python scripts/rocpd_summary.py $O/r03_prof_c8/c8_results.db $O/r03_config4_native_kernel_stats.md 2>&1 | head -16 >> $L
rm -rf $O/r03_prof_c8
cat $L
