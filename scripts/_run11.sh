export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests/test_solver_gpu.py tests/test_kernels_gpu.py tests/test_full_size_configs_gpu.py -x -q -p no:cacheprovider > $O/r02_t11.log 2>&1; tail -3 $O/r02_t11.log
python bench.py --no-cpu-baseline 2> $O/r02_b11.err | tail -1 > $O/r02_b11.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r02_b11.json'))
print('bench', d['value'], d['ms_per_step'], d['config']['us_per_outer_iteration'], d['roofline']['frac'], d['roofline']['all_kernels'])
ns=d.get('north_star') or {}
print('north', ns.get('seconds_per_solve'), ns.get('config',{}).get('us_per_outer_iteration'), ns.get('roofline',{}).get('inner_loop_all_classes'), ns.get('roofline',{}).get('spmv_plus_ortho'))
PY
PRIMME_AMD_NO_FUSED_RESTART=1 python bench.py --no-cpu-baseline --no-north-star 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('nofr', d['value'], d['ms_per_step'], d['config']['us_per_outer_iteration'])"
