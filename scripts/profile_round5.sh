#!/bin/bash
# Round 5 evidence pass (through gpurun): rocprofv3 kernel statistics + gaps of the WHOLE headline solve, the two PMC passes
# (FETCH_SIZE, WRITE_SIZE: separate runs, --kernel-trace only) over the WHOLE headline solve and over configs[1] — the same
# sample as bench.py's algorithmic bytes (VERDICT r04 Weak #10) —, kernel statistics of one solve of configs[2] / [3] / [4],
# and the request-size / hit counters of the gather probe.   usage: bash scripts/profile_round5.sh   (outputs: gpurun_out/r05_*)
R=$PWD; O=$R/gpurun_out; TAG=r05
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace -d $O/${TAG}_prof_ns -o ns -- python $R/scripts/one_solve.py csr lap2d_10m > $O/${TAG}_ns_run.log 2> $O/${TAG}_prof_ns.log; cat $O/${TAG}_ns_run.log
python $R/scripts/rocpd_summary.py $O/${TAG}_prof_ns/ns_results.db $O/${TAG}_headline_kernel_stats.md > /dev/null; head -14 $O/${TAG}_headline_kernel_stats.md; tail -1 $O/${TAG}_headline_kernel_stats.md
python $R/scripts/gap_analysis.py $O/${TAG}_prof_ns/ns_results.db $O/${TAG}_headline_gap_analysis.md > /dev/null; head -8 $O/${TAG}_headline_gap_analysis.md; tail -1 $O/${TAG}_headline_gap_analysis.md
rm -rf $O/${TAG}_prof_ns
cp $R/profiles/pmc_traffic.json $O/${TAG}_pmc_traffic.json
for wl in lap2d_10m lap3d_2m; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${TAG}_pmc_fetch -o p -- python $R/scripts/one_solve.py csr $wl > /dev/null 2> $O/${TAG}_pmc_fetch_$wl.log
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/${TAG}_pmc_write -o p -- python $R/scripts/one_solve.py csr $wl > /dev/null 2> $O/${TAG}_pmc_write_$wl.log
  python $R/scripts/pmc_traffic.py $O/${TAG}_pmc_fetch/p_results.db $O/${TAG}_pmc_write/p_results.db $O/${TAG}_pmc_traffic_$wl.md $O/${TAG}_pmc_traffic.json $R/profiles/r05_bench_line.json ${TAG} $wl | tail -3
  rm -rf $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write
done
cd $R
for c in 3 4 5; do
  ( cd /tmp && rocprofv3 --kernel-trace -d $O/${TAG}_prof_c$c -o c -- python $R/scripts/config${c}_run.py > $O/${TAG}_config${c}_run.log 2> $O/${TAG}_prof_c$c.log )
  python scripts/rocpd_summary.py $O/${TAG}_prof_c$c/c_results.db $O/${TAG}_config${c}_kernel_stats.md > /dev/null; tail -1 $O/${TAG}_config${c}_run.log | cut -c1-300; head -10 $O/${TAG}_config${c}_kernel_stats.md; tail -1 $O/${TAG}_config${c}_kernel_stats.md
  rm -rf $O/${TAG}_prof_c$c
done
# which L2 / fabric counters this box has, and the gather probe under the ones that say how large a miss request is
rocprofv3 -L 2>/dev/null | grep -o -E "TCC_EA0_RDREQ[A-Z0-9_]*|TCC_HIT[_a-z]*|TCC_MISS[_a-z]*|TCC_REQ[_a-z]*|TCC_READ[_a-z]*" | sort -u | tr '\n' ' ' > $O/${TAG}_tcc_counters_available.txt; cat $O/${TAG}_tcc_counters_available.txt; echo
( cd /tmp && rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $O/${TAG}_pmc_gp -o p -- $R/scripts/probes/gather_probe 2000000 > $O/${TAG}_pmc_gp.log 2>&1 )
DB=$(find $O/${TAG}_pmc_gp -name "*.db" | head -1); [ -n "$DB" ] && python scripts/pmc_summary.py $DB $O/${TAG}_gather_request_sizes.md; tail -3 $O/${TAG}_pmc_gp.log
rm -rf $O/${TAG}_pmc_gp
du -sh $O | tail -1
