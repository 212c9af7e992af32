#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_step6; mkdir -p $O
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider > $O/full_suite_a.txt 2>&1; echo "full suite (normal interpreter exit) exit code $?" | tee -a $O/full_suite_a.txt
tail -6 $O/full_suite_a.txt
cat gpurun_out/exact_history_gpu.json; echo
PRIMME_AMD_PRELAUNCH_STATS=1 timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; grep -v amdgpu.ids $O/bench.err | tail -4
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05_step6/bench.json"))
print("headline", d["value"], d["ms_per_step"], d["steps"], d["warmup"], d["config"]["us_per_outer_iteration"], d["roofline"]["frac"], d["roofline"]["achieved"])
print({k: (v["GBps"], v["ms"], v["launches"]) for k, v in d["roofline"]["all_kernels"].items()})
for k in ("configs1","configs2","configs3","configs4"):
    c=d.get(k,{}); print(k, c.get("value"), c.get("ms_per_step"), c.get("config",{}).get("us_per_outer_iteration"), c.get("roofline",{}).get("kernel","")[:40], c.get("roofline",{}).get("frac"), c.get("error"))
print(d.get("cpu_baseline"))
PY
