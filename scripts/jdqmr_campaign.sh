#!/bin/bash
# Round 3: block QMR with rho taken one pass early (hipk_axpy_proj_dot_jacobi + hipk_qmr_update_dir): kernel test, the
# JDQMR solver tests, configs[2] with and without, kernel stats
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
L=$O/r03_jdqmr_campaign.log
: > $L
timeout 600 python -X faulthandler -m pytest tests/test_kernels_gpu.py -k "qmr" -q -p no:cacheprovider > $O/r03_jd_kernels.log 2>&1; echo "rc=$?" >> $O/r03_jd_kernels.log
tail -4 $O/r03_jd_kernels.log >> $L
timeout 900 python -X faulthandler -m pytest tests/test_solver_gpu.py tests/test_full_size_configs_gpu.py -k "jdqmr or JDQMR or config3" -q -p no:cacheprovider > $O/r03_jd_solver.log 2>&1; echo "rc=$?" >> $O/r03_jd_solver.log
tail -4 $O/r03_jd_solver.log >> $L
echo "== configs[2] early rho" >> $L
timeout 300 python scripts/config3_run.py 2>&1 | tail -1 | cut -c1-330 >> $L
echo "== configs[2] round-2 sequence" >> $L
PRIMME_AMD_NO_EARLY_RHO=1 timeout 300 python scripts/config3_run.py 2>&1 | tail -1 | cut -c1-330 >> $L
timeout 300 rocprofv3 --kernel-trace -d $O/r03_prof_jd -o jd -- python scripts/config3_run.py > $O/r03_config3_run.log 2>&1
python scripts/rocpd_summary.py $O/r03_prof_jd/jd_results.db $O/r03_config3_kernel_stats.md 2>&1 | head -14 >> $L
rm -rf $O/r03_prof_jd
cat $L
