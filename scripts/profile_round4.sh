#!/bin/bash
# Round 4 evidence pass (through gpurun): the bench line, rocprofv3 kernel statistics + gaps of the headline workload
# (first 3000 outer iterations of lap2d_10m), and the two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, --kernel-trace only).
# usage: bash scripts/profile_round4.sh   (outputs: gpurun_out/r04_*)
O=$PWD/gpurun_out; TAG=r04
export TMPDIR=/tmp
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; tail -c 300 $O/${TAG}_bench.json; echo
rocprofv3 --kernel-trace -d $O/${TAG}_prof_ns -o ns -- python scripts/one_solve.py csr lap2d_10m 3000 > $O/${TAG}_ns_run.log 2> $O/${TAG}_prof_ns.log
python scripts/rocpd_summary.py $O/${TAG}_prof_ns/ns_results.db $O/${TAG}_headline_kernel_stats.md > /dev/null; head -12 $O/${TAG}_headline_kernel_stats.md; tail -1 $O/${TAG}_headline_kernel_stats.md
python scripts/gap_analysis.py $O/${TAG}_prof_ns/ns_results.db $O/${TAG}_headline_gap_analysis.md > /dev/null; tail -1 $O/${TAG}_headline_gap_analysis.md
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${TAG}_pmc_fetch_ns -o p -- python scripts/one_solve.py csr lap2d_10m 400 > /dev/null 2> $O/${TAG}_pmc_fetch_ns.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/${TAG}_pmc_write_ns -o p -- python scripts/one_solve.py csr lap2d_10m 400 > /dev/null 2> $O/${TAG}_pmc_write_ns.log
cp profiles/pmc_traffic.json $O/${TAG}_pmc_traffic.json
python scripts/pmc_traffic.py $O/${TAG}_pmc_fetch_ns/p_results.db $O/${TAG}_pmc_write_ns/p_results.db $O/${TAG}_pmc_traffic_lap2d_10m.md $O/${TAG}_pmc_traffic.json $O/${TAG}_bench.json ${TAG} lap2d_10m | tail -4
rm -rf $O/${TAG}_prof_ns $O/${TAG}_pmc_fetch_ns $O/${TAG}_pmc_write_ns
