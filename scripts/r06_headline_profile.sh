# kernel trace of ONE headline solve (lap2d_10m, 10 smallest, GD+k): per-kernel table and idle-gap table for profiles/
R=$PWD; O=$R/gpurun_out; TAG=r06
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace -d $O/${TAG}_prof_head -o h -- python $R/scripts/one_solve.py csr lap2d_10m > $O/${TAG}_headline_run.log 2> $O/${TAG}_prof_head.log
cat $O/${TAG}_headline_run.log
DB=$(find $O/${TAG}_prof_head -name "*.db" | head -1)
python $R/scripts/rocpd_summary.py $DB $O/${TAG}_headline_kernel_stats.md > /dev/null; head -16 $O/${TAG}_headline_kernel_stats.md; tail -1 $O/${TAG}_headline_kernel_stats.md
python $R/scripts/gap_analysis.py $DB $O/${TAG}_headline_gap_analysis.md > /dev/null; head -10 $O/${TAG}_headline_gap_analysis.md; tail -1 $O/${TAG}_headline_gap_analysis.md
rm -rf $O/${TAG}_prof_head
