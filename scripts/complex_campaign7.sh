#!/bin/bash
# Round 3, seventh GPU pass: copy kernel for small pinned transfers, U-column load batches in the multi-lane restart kernel
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
L=$O/r03_complex_campaign7.log
: > $L
echo "== kernel tests (complex + real ritz/core)" >> $L
timeout 900 python -X faulthandler -m pytest tests/test_kernels_complex_gpu.py tests/test_kernels_gpu.py -q -p no:cacheprovider -x > $O/r03_c7_kernels.log 2>&1; echo "rc=$?" >> $O/r03_c7_kernels.log
tail -4 $O/r03_c7_kernels.log >> $L
echo "== solver tests" >> $L
timeout 900 python -X faulthandler -m pytest tests/test_complex_gpu.py tests/test_solver_gpu.py -q -p no:cacheprovider > $O/r03_c7_solver.log 2>&1; echo "rc=$?" >> $O/r03_c7_solver.log
tail -4 $O/r03_c7_solver.log >> $L
echo "== configs[3] native" >> $L
FORM=native timeout 300 python scripts/config4_run.py 2>&1 | cut -c1-120 >> $L
HIPK_NO_COPY_KERNEL=1 FORM=native timeout 300 python scripts/config4_run.py 2>&1 | cut -c1-120 >> $L
FORM=native timeout 300 rocprofv3 --kernel-trace -d $O/r03_prof_c7 -o c7 -- python scripts/config4_run.py > $O/r03_config4_under_rocprof.log 2>&1
python scripts/rocpd_gaps.py $O/r03_prof_c7/c7_results.db $O/r03_config4_native_gaps.md 2>&1 | head -16 >> $L
python scripts/rocpd_summary.py $O/r03_prof_c7/c7_results.db $O/r03_config4_native_kernel_stats.md 2>&1 | head -16 >> $L
rm -rf $O/r03_prof_c7
echo "== bench" >> $L
timeout 600 python bench.py 2>&1 | tail -1 | cut -c1-400 >> $L
HIPK_NO_COPY_KERNEL=1 timeout 600 python bench.py 2>&1 | tail -1 | cut -c1-200 >> $L
cat $L
