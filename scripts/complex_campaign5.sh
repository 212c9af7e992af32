#!/bin/bash
# Round 3, fifth GPU pass: complex harmonic / refined fixtures on the GPU, and where the device idles in configs[3]
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
L=$O/r03_complex_campaign5.log
: > $L
echo "== complex fixtures + refined/harmonic real tests" >> $L
timeout 600 python -X faulthandler -m pytest tests/test_complex_gpu.py tests/test_solver_gpu.py -k "fixture or projection or refined or harmonic" -q -p no:cacheprovider > $O/r03_c5_fixtures.log 2>&1; echo "rc=$?" >> $O/r03_c5_fixtures.log
tail -6 $O/r03_c5_fixtures.log >> $L
echo "== configs[3] native under rocprof: idle gaps" >> $L
FORM=native timeout 300 rocprofv3 --kernel-trace -d $O/r03_prof_c5 -o c5 -- python scripts/config4_run.py > $O/r03_config4_under_rocprof.log 2>&1
python scripts/rocpd_gaps.py $O/r03_prof_c5/c5_results.db $O/r03_config4_native_gaps.md >> $L 2>&1
python scripts/rocpd_summary.py $O/r03_prof_c5/c5_results.db $O/r03_config4_native_kernel_stats.md > /dev/null 2>&1
rm -rf $O/r03_prof_c5
cat $L
