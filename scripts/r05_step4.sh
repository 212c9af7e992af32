#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_step4; mkdir -p $O
export PRIMME_AMD_PRELAUNCH_STATS=1
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_reference_kernels.py -m gpu -q -p no:cacheprovider -k "rr_arrow or pattern or csr or reference_routines" > $O/tests_k.txt 2>&1; echo "kernel tests exit $?"; tail -4 $O/tests_k.txt
timeout 900 python -m pytest tests/test_solver_gpu.py tests/test_svds_gpu.py tests/test_interface_cases_gpu.py -m gpu -q -p no:cacheprovider > $O/tests_s.txt 2>&1; echo "solver tests exit $?"; grep -v "enqueued ahead" $O/tests_s.txt | tail -6; grep -c "enqueued ahead" $O/tests_s.txt
cat gpurun_out/exact_history_gpu.json; echo
for nb in 0 1; do
  if [ $nb = 1 ]; then export PRIMME_AMD_NO_PRELAUNCH=1; else unset PRIMME_AMD_NO_PRELAUNCH; fi
  timeout 600 python bench.py --workload lap3d_2m --steps 3 --warmup 1 --no-extra-configs --no-cpu-baseline > $O/bench_c1_noprelaunch$nb.json 2> $O/bench_c1_noprelaunch$nb.err
  echo "configs1 NO_PRELAUNCH=$nb: $(python -c "import json;d=json.load(open('$O/bench_c1_noprelaunch$nb.json'));print(d['value'], d['ms_per_step'], d['config']['us_per_outer_iteration'], d['config']['outer_iterations'])")"; grep "enqueued ahead" $O/bench_c1_noprelaunch$nb.err | tail -1
done
unset PRIMME_AMD_NO_PRELAUNCH
timeout 900 python bench.py --steps 1 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; grep -v amdgpu.ids $O/bench.err | tail -5
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05_step4/bench.json"))
print("headline", d["value"], d["ms_per_step"], d["config"]["us_per_outer_iteration"], d["roofline"]["frac"])
for k in ("configs1","configs2","configs3","configs4"):
    c=d.get(k,{}); print(k, c.get("value"), c.get("ms_per_step"), c.get("config",{}).get("us_per_outer_iteration"), c.get("roofline",{}).get("kernel","")[:40], c.get("roofline",{}).get("frac"), c.get("error"))
PY
