#!/bin/bash
# Round 3, fourth GPU pass: two-lanes-per-row complex restart kernel (zritz2), complex solver tests,
# configs[3] native / real form timing, kernel stats of the native run.
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
L=$O/r03_complex_campaign4.log
: > $L
echo "== complex kernel tests + larnv" >> $L
timeout 600 python -X faulthandler -m pytest tests/test_kernels_complex_gpu.py tests/test_kernels_gpu.py -k "complex or larnv" -q -p no:cacheprovider > $O/r03_c4_kernels.log 2>&1; echo "rc=$?" >> $O/r03_c4_kernels.log
tail -6 $O/r03_c4_kernels.log >> $L
echo "== complex solver tests" >> $L
timeout 600 python -X faulthandler -m pytest tests/test_complex_gpu.py -q -p no:cacheprovider > $O/r03_c4_solver.log 2>&1; echo "rc=$?" >> $O/r03_c4_solver.log
tail -6 $O/r03_c4_solver.log >> $L
echo "== configs[3] native / real form" >> $L
HIPK_HOST_TIMING=1 FORM=native timeout 300 python scripts/config4_run.py 2>&1 | cut -c1-200 >> $L
FORM=real timeout 300 python scripts/config4_run.py 2>&1 | cut -c1-200 >> $L
FORM=native timeout 300 rocprofv3 --kernel-trace -d $O/r03_prof_c4 -o c4 -- python scripts/config4_run.py > $O/r03_config4_under_rocprof.log 2>&1
python scripts/rocpd_summary.py $O/r03_prof_c4/c4_results.db $O/r03_config4_native_kernel_stats.md > /dev/null 2>&1; head -22 $O/r03_config4_native_kernel_stats.md >> $L; tail -1 $O/r03_config4_native_kernel_stats.md >> $L
rm -rf $O/r03_prof_c4
cat $L
