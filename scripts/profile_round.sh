#!/bin/bash
# One GPU-box pass that produces everything profiles/ is built from (run through gpurun):
#   full -m gpu suite, bench line, rocprofv3 kernel stats of the bench and of configs[2], PMC passes
#   (MFMA counters on a block solve; FETCH_SIZE / WRITE_SIZE on the bench solve).
# usage: bash scripts/profile_round.sh <tag>     (outputs: gpurun_out/<tag>_*)
TAG=${1:-r03}
O=gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -p no:cacheprovider > $O/${TAG}_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_gpu_tests.log; tail -4 $O/${TAG}_gpu_tests.log
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; tail -c 400 $O/${TAG}_bench.json; echo
rocprofv3 --kernel-trace -d $O/${TAG}_prof_bench -o bench -- python bench.py --no-cpu-baseline --no-north-star > $O/${TAG}_bench_under_rocprof.json 2> $O/${TAG}_prof_bench.log
python scripts/rocpd_summary.py $O/${TAG}_prof_bench/bench_results.db $O/${TAG}_bench_kernel_stats.md > /dev/null; head -14 $O/${TAG}_bench_kernel_stats.md; tail -1 $O/${TAG}_bench_kernel_stats.md
python scripts/gap_analysis.py $O/${TAG}_prof_bench/bench_results.db $O/${TAG}_bench_gap_analysis.md > /dev/null; tail -1 $O/${TAG}_bench_gap_analysis.md
rocprofv3 --kernel-trace -d $O/${TAG}_prof_c3 -o c3 -- python scripts/config3_run.py --prof > $O/${TAG}_config3_run.log 2> $O/${TAG}_prof_c3.log
python scripts/rocpd_summary.py $O/${TAG}_prof_c3/c3_results.db $O/${TAG}_config3_kernel_stats.md > /dev/null; tail -2 $O/${TAG}_config3_run.log | cut -c1-600; head -16 $O/${TAG}_config3_kernel_stats.md; tail -1 $O/${TAG}_config3_kernel_stats.md
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -d $O/${TAG}_pmc_mfma -o p -- python scripts/config3_run.py --tiles 8000 > $O/${TAG}_pmc_mfma_run.log 2> $O/${TAG}_pmc_mfma.log
python scripts/pmc_summary.py $O/${TAG}_pmc_mfma/p_results.db $O/${TAG}_pmc_mfma.md > /dev/null; head -12 $O/${TAG}_pmc_mfma.md
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${TAG}_pmc_fetch -o p -- python scripts/one_solve.py csr > /dev/null 2> $O/${TAG}_pmc_fetch.log
python scripts/pmc_summary.py $O/${TAG}_pmc_fetch/p_results.db $O/${TAG}_pmc_fetch.md > /dev/null
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/${TAG}_pmc_write -o p -- python scripts/one_solve.py csr > /dev/null 2> $O/${TAG}_pmc_write.log
python scripts/pmc_summary.py $O/${TAG}_pmc_write/p_results.db $O/${TAG}_pmc_write.md > /dev/null
python scripts/pmc_traffic.py $O/${TAG}_pmc_fetch/p_results.db $O/${TAG}_pmc_write/p_results.db $O/${TAG}_pmc_traffic.md $O/${TAG}_pmc_traffic.json $O/${TAG}_bench.json ${TAG}
# ---- configs[4] SpMV pair: how many bytes the two products really move (each 8-byte gather of this matrix pulls its own
#      cache line: the kernels are bound by lines, not by the algorithmic bytes)
python scripts/config5_spmv_perf.py > $O/${TAG}_config5_spmv_pair.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${TAG}_pmc_c5 -o p -- python scripts/config5_spmv_perf.py > /dev/null 2> $O/${TAG}_pmc_c5.log
python scripts/pmc_summary.py $O/${TAG}_pmc_c5/p_results.db $O/${TAG}_pmc_config5_spmv_fetch.md > /dev/null; cat $O/${TAG}_pmc_config5_spmv_fetch.md | head -6
rm -rf $O/${TAG}_pmc_c5
# ---- north-star workload (10 M-row 5-pt Laplacian): kernel statistics of the first 3000 outer iterations and the two PMC passes
rocprofv3 --kernel-trace -d $O/${TAG}_prof_ns -o ns -- python scripts/one_solve.py csr lap2d_10m 3000 > $O/${TAG}_ns_run.log 2> $O/${TAG}_prof_ns.log
python scripts/rocpd_summary.py $O/${TAG}_prof_ns/ns_results.db $O/${TAG}_north_star_kernel_stats.md > /dev/null; head -12 $O/${TAG}_north_star_kernel_stats.md; tail -1 $O/${TAG}_north_star_kernel_stats.md
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${TAG}_pmc_fetch_ns -o p -- python scripts/one_solve.py csr lap2d_10m 400 > /dev/null 2> $O/${TAG}_pmc_fetch_ns.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/${TAG}_pmc_write_ns -o p -- python scripts/one_solve.py csr lap2d_10m 400 > /dev/null 2> $O/${TAG}_pmc_write_ns.log
cp $O/${TAG}_pmc_traffic.json $O/${TAG}_pmc_traffic_all.json 2>/dev/null
python scripts/pmc_traffic.py $O/${TAG}_pmc_fetch_ns/p_results.db $O/${TAG}_pmc_write_ns/p_results.db $O/${TAG}_pmc_traffic_lap2d_10m.md $O/${TAG}_pmc_traffic.json $O/${TAG}_bench.json ${TAG} lap2d_10m
rm -rf $O/${TAG}_prof_ns $O/${TAG}_pmc_fetch_ns $O/${TAG}_pmc_write_ns
rm -rf $O/${TAG}_prof_bench $O/${TAG}_prof_c3 $O/${TAG}_pmc_mfma $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write
