#!/bin/bash
# Round 6, step 2 (through gpurun): tests of the new fused kernels, kernel trace + gaps of configs[1] with the one-launch tail,
# A/B timings of configs[2] (project + triple dots in one pass) and configs[3] (complex TN panel on the matrix cores).
R=$PWD; O=$R/gpurun_out; TAG=r06
export TMPDIR=/tmp
python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "project_and_triple or tail_without" 2>&1 | tail -4 > $O/${TAG}_s2_tests.log
python -m pytest tests/test_kernels_complex_gpu.py tests/test_complex_gpu.py -q -m gpu -x 2>&1 | tail -6 >> $O/${TAG}_s2_tests.log
python -m pytest tests/test_solver_gpu.py -q -m gpu -x -k "jd or JD or block or blk" 2>&1 | tail -4 >> $O/${TAG}_s2_tests.log
cat $O/${TAG}_s2_tests.log
cd /tmp
for v in defer nodefer; do
  if [ $v = nodefer ]; then export HIPK_NO_TAIL_DEFER=1; else unset HIPK_NO_TAIL_DEFER; fi
  rocprofv3 --kernel-trace -d $O/${TAG}_prof_c1_$v -o c1 -- python $R/scripts/one_solve.py csr lap3d_2m > $O/${TAG}_c1_${v}_run.log 2> $O/${TAG}_prof_c1_$v.log; cat $O/${TAG}_c1_${v}_run.log
  python $R/scripts/rocpd_summary.py $O/${TAG}_prof_c1_$v/c1_results.db $O/${TAG}_configs1_kernel_stats_$v.md > /dev/null; head -14 $O/${TAG}_configs1_kernel_stats_$v.md; tail -1 $O/${TAG}_configs1_kernel_stats_$v.md
  python $R/scripts/gap_analysis.py $O/${TAG}_prof_c1_$v/c1_results.db $O/${TAG}_configs1_gap_analysis_$v.md > /dev/null; head -12 $O/${TAG}_configs1_gap_analysis_$v.md; tail -1 $O/${TAG}_configs1_gap_analysis_$v.md
  rm -rf $O/${TAG}_prof_c1_$v
done
unset HIPK_NO_TAIL_DEFER
cd $R
for v in fused pair; do
  if [ $v = pair ]; then export PRIMME_AMD_NO_PROJECT_TRIPLE=1; else unset PRIMME_AMD_NO_PROJECT_TRIPLE; fi
  python scripts/config3_run.py --reps 3 2>&1 | tail -2 | cut -c1-400 > $O/${TAG}_config3_$v.log; cat $O/${TAG}_config3_$v.log
done
unset PRIMME_AMD_NO_PROJECT_TRIPLE
for v in mfma valu; do
  if [ $v = valu ]; then export HIPK_NO_ZMFMA=1; else unset HIPK_NO_ZMFMA; fi
  python scripts/config4_run.py 2>&1 | cut -c1-330 > $O/${TAG}_config4_$v.log; cat $O/${TAG}_config4_$v.log
done
