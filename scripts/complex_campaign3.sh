#!/bin/bash
# Round 3, third GPU pass: column-split complex TN kernel, device LARNV, repeated grouped runs (hunting the one core dump
# of the first pass), configs[3] timing, full suite.
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
L=$O/r03_complex_campaign3.log
: > $L
echo "== complex kernel tests + larnv" >> $L
timeout 600 python -X faulthandler -m pytest tests/test_kernels_complex_gpu.py tests/test_kernels_gpu.py -k "complex or larnv" -q -p no:cacheprovider > $O/r03_c3_kernels.log 2>&1; echo "rc=$?" >> $O/r03_c3_kernels.log
tail -6 $O/r03_c3_kernels.log >> $L
for i in 1 2 3 4; do
  echo "== grouped run $i" >> $L
  timeout 600 python -X faulthandler -m pytest tests/test_complex_gpu.py tests/test_c_examples_gpu.py tests/test_kernels_gpu.py -q -p no:cacheprovider > $O/r03_c3_group$i.log 2>&1; echo "rc=$?" >> $O/r03_c3_group$i.log
  tail -4 $O/r03_c3_group$i.log >> $L
  grep -n -i "fatal\|fault\|Segmentation\|Abort\|core" $O/r03_c3_group$i.log | head -5 >> $L
done
echo "== configs[3] native / real form" >> $L
HIPK_HOST_TIMING=1 FORM=native timeout 300 python scripts/config4_run.py 2>&1 | cut -c1-200 >> $L
FORM=real timeout 300 python scripts/config4_run.py 2>&1 | cut -c1-200 >> $L
FORM=native timeout 300 rocprofv3 --kernel-trace -d $O/r03_prof_c4 -o c4 -- python scripts/config4_run.py > $O/r03_config4_under_rocprof.log 2>&1
python scripts/rocpd_summary.py $O/r03_prof_c4/c4_results.db $O/r03_config4_native_kernel_stats.md > /dev/null 2>&1; head -22 $O/r03_config4_native_kernel_stats.md >> $L; tail -1 $O/r03_config4_native_kernel_stats.md >> $L
rm -rf $O/r03_prof_c4
echo "== full GPU suite" >> $L
timeout 1200 python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider > $O/r03_gpu_suite_run3.log 2>&1; echo "pytest rc=$?" >> $O/r03_gpu_suite_run3.log
tail -12 $O/r03_gpu_suite_run3.log >> $L
cat $L
