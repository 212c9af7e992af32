#!/bin/bash
# Round 3, sixth GPU pass: four-lanes-per-row restart kernel (up to 64 outputs in one pass): kernel tests, complex solver
# tests, configs[3] timing and idle gaps
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
L=$O/r03_complex_campaign6.log
: > $L
echo "== complex kernel tests" >> $L
timeout 600 python -X faulthandler -m pytest tests/test_kernels_complex_gpu.py -q -p no:cacheprovider > $O/r03_c6_kernels.log 2>&1; echo "rc=$?" >> $O/r03_c6_kernels.log
tail -6 $O/r03_c6_kernels.log >> $L
echo "== complex solver tests" >> $L
timeout 600 python -X faulthandler -m pytest tests/test_complex_gpu.py -q -p no:cacheprovider > $O/r03_c6_solver.log 2>&1; echo "rc=$?" >> $O/r03_c6_solver.log
tail -6 $O/r03_c6_solver.log >> $L
echo "== configs[3] native / real form" >> $L
HIPK_HOST_TIMING=1 FORM=native timeout 300 python scripts/config4_run.py 2>&1 | cut -c1-200 >> $L
FORM=native timeout 300 rocprofv3 --kernel-trace -d $O/r03_prof_c6 -o c6 -- python scripts/config4_run.py > $O/r03_config4_under_rocprof.log 2>&1
python scripts/rocpd_gaps.py $O/r03_prof_c6/c6_results.db $O/r03_config4_native_gaps.md 2>&1 | head -30 >> $L
python scripts/rocpd_summary.py $O/r03_prof_c6/c6_results.db $O/r03_config4_native_kernel_stats.md 2>&1 | head -24 >> $L
rm -rf $O/r03_prof_c6
cat $L
