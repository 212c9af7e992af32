export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/r02_t18.log 2>&1; tail -3 $O/r02_t18.log
python scripts/config3_run.py 2>&1 | tail -1 | cut -c1-220
python bench.py --no-cpu-baseline --no-north-star 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['config']['us_per_outer_iteration'])"
