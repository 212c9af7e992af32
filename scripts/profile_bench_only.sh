#!/bin/bash
# the bench part of scripts/profile_round.sh: bench line, rocprofv3 kernel stats + gap table of the bench, PMC traffic of
# configs[1] and (first 400 iterations) of the north-star workload.   usage: bash scripts/profile_bench_only.sh <tag>
TAG=${1:-r03n}
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; tail -c 300 $O/${TAG}_bench.json; echo
rocprofv3 --kernel-trace -d $O/${TAG}_prof_bench -o bench -- python bench.py --no-cpu-baseline --no-north-star > $O/${TAG}_bench_under_rocprof.json 2> $O/${TAG}_prof_bench.log
python scripts/rocpd_summary.py $O/${TAG}_prof_bench/bench_results.db $O/${TAG}_bench_kernel_stats.md > /dev/null; head -12 $O/${TAG}_bench_kernel_stats.md; tail -1 $O/${TAG}_bench_kernel_stats.md
python scripts/gap_analysis.py $O/${TAG}_prof_bench/bench_results.db $O/${TAG}_bench_gap_analysis.md > /dev/null; tail -1 $O/${TAG}_bench_gap_analysis.md
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${TAG}_pmc_fetch -o p -- python scripts/one_solve.py csr > /dev/null 2> $O/${TAG}_pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/${TAG}_pmc_write -o p -- python scripts/one_solve.py csr > /dev/null 2> $O/${TAG}_pmc_write.log
python scripts/pmc_traffic.py $O/${TAG}_pmc_fetch/p_results.db $O/${TAG}_pmc_write/p_results.db $O/${TAG}_pmc_traffic.md $O/${TAG}_pmc_traffic.json $O/${TAG}_bench.json ${TAG}
rocprofv3 --kernel-trace -d $O/${TAG}_prof_ns -o ns -- python scripts/one_solve.py csr lap2d_10m 3000 > $O/${TAG}_ns_run.log 2> $O/${TAG}_prof_ns.log
python scripts/rocpd_summary.py $O/${TAG}_prof_ns/ns_results.db $O/${TAG}_north_star_kernel_stats.md > /dev/null; head -8 $O/${TAG}_north_star_kernel_stats.md
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${TAG}_pmc_fetch_ns -o p -- python scripts/one_solve.py csr lap2d_10m 400 > /dev/null 2> $O/${TAG}_pmc_fetch_ns.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/${TAG}_pmc_write_ns -o p -- python scripts/one_solve.py csr lap2d_10m 400 > /dev/null 2> $O/${TAG}_pmc_write_ns.log
python scripts/pmc_traffic.py $O/${TAG}_pmc_fetch_ns/p_results.db $O/${TAG}_pmc_write_ns/p_results.db $O/${TAG}_pmc_traffic_lap2d_10m.md $O/${TAG}_pmc_traffic.json $O/${TAG}_bench.json ${TAG} lap2d_10m
rm -rf $O/${TAG}_prof_bench $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write $O/${TAG}_prof_ns $O/${TAG}_pmc_fetch_ns $O/${TAG}_pmc_write_ns
