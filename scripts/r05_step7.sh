#!/bin/bash
# round 5, step 7: the pre-enqueued iteration on the mailbox transport (ranks as processes on the one GPU of the box)
cd "$(dirname "$0")/.."
O=gpurun_out/r05_step7; mkdir -p $O
timeout 1200 python -m pytest tests/test_multirank_ipc_gpu.py tests/test_comm_gpu.py tests/test_multigpu_rccl.py -m gpu -x -q -p no:cacheprovider > $O/multirank.txt 2>&1; echo "multirank tests exit $?" | tee -a $O/multirank.txt
tail -15 $O/multirank.txt
B="--workload lap3d_2m --steps 3 --warmup 1 --no-configs1 --no-extra-configs --no-cpu-baseline"
for N in 1 2 4; do
  for PRE in on off; do
    if [ $PRE = off ]; then export PRIMME_AMD_NO_PRELAUNCH=1; else unset PRIMME_AMD_NO_PRELAUNCH; fi
    PRIMME_AMD_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus $N $B > $O/bench_n${N}_$PRE.json 2> $O/bench_n${N}_$PRE.err
    echo "N=$N prelaunch=$PRE exit $?"
    python - $O/bench_n${N}_$PRE.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    c = d["config"]
    print("  ", d["value"], "eig/s", c["us_per_outer_iteration"], "us/iter", c["outer_iterations"], "its", c.get("iterations_enqueued_ahead"), c.get("transport"), c.get("comm_selftest", {}).get("allreduce_us"))
except Exception as e:
    print("  no line:", e)
PY
  done
done
unset PRIMME_AMD_NO_PRELAUNCH
grep -v amdgpu.ids $O/bench_n2_on.err | tail -5
