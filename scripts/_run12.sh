export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_solver_gpu.py -x -q -p no:cacheprovider > $O/r02_t12.log 2>&1; tail -2 $O/r02_t12.log
python scripts/kernel_perf.py 2>&1 | tail -25
python bench.py --no-cpu-baseline --no-north-star 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['config']['us_per_outer_iteration'], d['roofline']['frac'], d['roofline']['all_kernels'])"
