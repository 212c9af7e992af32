export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_solver_gpu.py -x -q -p no:cacheprovider > $O/r02_t10.log 2>&1; tail -2 $O/r02_t10.log
for m in 0 2 1 3; do
  HIPK_HOST_TIMING=1 HIPK_INKERNEL_FIN=$m python bench.py --no-cpu-baseline --no-north-star 2> $O/r02_b10_$m.err | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('mask $m', d['value'], d['ms_per_step'], d['config']['us_per_outer_iteration'], d['roofline']['frac'])"
  grep "host timing" $O/r02_b10_$m.err | tail -2
done
rocprofv3 --kernel-trace -d $O/r02_gap -o bench -- python bench.py --no-cpu-baseline --no-north-star > /dev/null 2> $O/r02_gap.log
python scripts/gap_analysis.py $O/r02_gap/bench_results.db $O/r02_gap_analysis.md | cut -c1-160 | head -16
python scripts/rocpd_summary.py $O/r02_gap/bench_results.db | grep -i "finalize\|busy"
rm -rf $O/r02_gap
