#!/bin/bash
# Round 3: first GPU pass of the complex instantiation (kernels, native hip_zprimme, configs[3] A/B) + full suite.
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
L=$O/r03_complex_campaign.log
: > $L
echo "== complex kernel tests" >> $L
timeout 600 python -m pytest tests/test_kernels_complex_gpu.py -q -p no:cacheprovider -x 2>&1 | tail -25 >> $L
echo "== complex solver tests + C examples + real kernel tests (new harness)" >> $L
timeout 900 python -m pytest tests/test_complex_gpu.py tests/test_c_examples_gpu.py tests/test_kernels_gpu.py -q -p no:cacheprovider 2>&1 | tail -25 >> $L
echo "== configs[3], native complex vs real-equivalent form" >> $L
FORM=native timeout 300 python scripts/config4_run.py >> $L 2>&1
FORM=real timeout 300 python scripts/config4_run.py >> $L 2>&1
echo "== rocprof of the native run" >> $L
FORM=native timeout 300 rocprofv3 --kernel-trace -d $O/r03_prof_c4 -o c4 -- python scripts/config4_run.py > $O/r03_config4_under_rocprof.log 2>&1
python scripts/rocpd_summary.py $O/r03_prof_c4/c4_results.db $O/r03_config4_native_kernel_stats.md > /dev/null 2>&1; head -30 $O/r03_config4_native_kernel_stats.md >> $L; tail -2 $O/r03_config4_native_kernel_stats.md >> $L
rm -rf $O/r03_prof_c4
echo "== full GPU suite" >> $L
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r03_gpu_suite_run2.log 2>&1; echo "pytest rc=$?" >> $O/r03_gpu_suite_run2.log
tail -15 $O/r03_gpu_suite_run2.log >> $L
cat $L
