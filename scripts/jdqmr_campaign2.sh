#!/bin/bash
# Round 3: pair loads in the two QMR passes: kernel test, configs[2] A/B, kernel stats
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
L=$O/r03_jdqmr_campaign2.log
: > $L
timeout 600 python -X faulthandler -m pytest tests/test_kernels_gpu.py -k "qmr" -q -p no:cacheprovider > $O/r03_jd2_kernels.log 2>&1; echo "rc=$?" >> $O/r03_jd2_kernels.log
tail -3 $O/r03_jd2_kernels.log >> $L
echo "== configs[2] pair loads" >> $L
timeout 300 python scripts/config3_run.py 2>&1 | tail -1 | cut -c1-330 >> $L
echo "== configs[2] HIPK_NO_PAIR_LOADS=1" >> $L
HIPK_NO_PAIR_LOADS=1 timeout 300 python scripts/config3_run.py 2>&1 | tail -1 | cut -c1-120 >> $L
timeout 300 rocprofv3 --kernel-trace -d $O/r03_prof_jd -o jd -- python scripts/config3_run.py > $O/r03_config3_run.log 2>&1
python scripts/rocpd_summary.py $O/r03_prof_jd/jd_results.db $O/r03_config3_kernel_stats.md 2>&1 | head -12 >> $L
rm -rf $O/r03_prof_jd
cat $L
