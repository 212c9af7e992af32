# configs[1] (lap3d_2m) with the rows over N processes on the ONE MI355X of the box (functional mode): us per outer iteration,
# default tail (round 6) against HIPK_NO_TAIL_DEFER=1 (round 5's sequence of launches)
R=$PWD; O=$R/gpurun_out
export PRIMME_AMD_BENCH_SHARE_GPU=1
for n in 1 2 4; do for v in defer nodefer; do
  if [ $v = nodefer ]; then export HIPK_NO_TAIL_DEFER=1; else unset HIPK_NO_TAIL_DEFER; fi
  python bench.py --gpus $n --workload lap3d_2m --steps 3 --warmup 1 --no-extra-configs --no-cpu-baseline --no-configs1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['config']
print('$n', '$v', d['value'], c['us_per_outer_iteration'], c['outer_iterations'], c.get('iterations_enqueued_ahead'), c.get('transport'), (c.get('comm_selftest') or {}).get('allreduce_us'))"
done; done
