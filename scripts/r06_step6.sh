#!/bin/bash
# Round 6, step 6: the SpMM window's odd LDS stride (tests + A/B on configs[2]), MFMA counters of configs[2] / configs[3],
# HBM traffic of the final build (FETCH_SIZE / WRITE_SIZE passes) for the headline and configs[1].
R=$PWD; O=$R/gpurun_out; TAG=r06
export TMPDIR=/tmp
python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "csr" 2>&1 | tail -3 > $O/${TAG}_s6_tests.log; cat $O/${TAG}_s6_tests.log
( for i in 1 2; do for v in odd even; do if [ $v = even ]; then export HIPK_SPMM_EVEN_STRIDE=1; else unset HIPK_SPMM_EVEN_STRIDE; fi
  echo "window stride $v"; python scripts/config3_run.py --reps 3 2>&1 | tail -1 | cut -c1-140; done; done ) > $O/${TAG}_spmm_window_stride.txt 2>&1
unset HIPK_SPMM_EVEN_STRIDE; cat $O/${TAG}_spmm_window_stride.txt
cd /tmp
rocprofv3 --kernel-trace -d $O/${TAG}_prof_c3 -o c -- python $R/scripts/config3_run.py > $O/${TAG}_config3_run.log 2> $O/${TAG}_prof_c3.log
python $R/scripts/rocpd_summary.py $O/${TAG}_prof_c3/c_results.db $O/${TAG}_config3_kernel_stats.md > /dev/null; head -12 $O/${TAG}_config3_kernel_stats.md; tail -1 $O/${TAG}_config3_kernel_stats.md
rm -rf $O/${TAG}_prof_c3
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace -d $O/${TAG}_pmc_m2 -o p -- python $R/scripts/config3_run.py > /dev/null 2> $O/${TAG}_pmc_m2.log
DB=$(find $O/${TAG}_pmc_m2 -name "*.db" | head -1); [ -n "$DB" ] && python $R/scripts/pmc_mfma.py $DB $O/${TAG}_pmc_mfma_configs2.md $O/${TAG}_pmc_mfma.json configs2
rm -rf $O/${TAG}_pmc_m2
HIPK_ZMFMA=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace -d $O/${TAG}_pmc_m3 -o p -- python $R/scripts/config4_run.py > /dev/null 2> $O/${TAG}_pmc_m3.log
DB=$(find $O/${TAG}_pmc_m3 -name "*.db" | head -1); [ -n "$DB" ] && python $R/scripts/pmc_mfma.py $DB $O/${TAG}_pmc_mfma_configs3_zmfma.md $O/${TAG}_pmc_mfma.json configs3
rm -rf $O/${TAG}_pmc_m3
cp $R/profiles/pmc_traffic.json $O/${TAG}_pmc_traffic.json
for wl in lap3d_2m lap2d_10m; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${TAG}_pmc_fetch -o p -- python $R/scripts/one_solve.py csr $wl > /dev/null 2> $O/${TAG}_pmc_fetch_$wl.log
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/${TAG}_pmc_write -o p -- python $R/scripts/one_solve.py csr $wl > /dev/null 2> $O/${TAG}_pmc_write_$wl.log
  python $R/scripts/pmc_traffic.py $O/${TAG}_pmc_fetch/p_results.db $O/${TAG}_pmc_write/p_results.db $O/${TAG}_pmc_traffic_$wl.md $O/${TAG}_pmc_traffic.json $O/${TAG}_bench2.json ${TAG} $wl | tail -3
  rm -rf $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write
done
du -sh $O | tail -1
