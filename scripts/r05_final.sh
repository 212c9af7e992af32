#!/bin/bash
# round 5, final evidence pass: full GPU suite (normal interpreter exit, run 5 of the round), bench.py with default flags, kernel
# statistics + gaps of one headline solve and of one configs[1] solve under rocprofv3 --kernel-trace
R=$PWD; O=$R/gpurun_out/r05_final; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider > $O/full_suite.txt 2>&1; echo "full suite (normal interpreter exit) exit code $?" | tee -a $O/full_suite.txt
tail -4 $O/full_suite.txt
cp gpurun_out/exact_history_gpu.json $O/ 2>/dev/null
PRIMME_AMD_PRELAUNCH_STATS=1 timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; grep -v amdgpu.ids $O/bench.err | tail -3
python - <<'PY'
import json
d=[json.loads(l) for l in open("gpurun_out/r05_final/bench.json") if l.startswith('{"metric"')][0]
print("headline", d["value"], d["ms_per_step"], d["steps"], d["warmup"], d["config"]["us_per_outer_iteration"], d["roofline"]["frac"], d["roofline"]["achieved"], d["config"].get("iterations_enqueued_ahead"))
print({k: (v["GBps"], v["ms"], v["launches"]) for k, v in d["roofline"]["all_kernels"].items()})
for k in ("configs1","configs2","configs3","configs4"):
    c=d.get(k,{}); print(k, c.get("value"), c.get("ms_per_step"), c.get("config",{}).get("us_per_outer_iteration"), c.get("roofline",{}).get("kernel","")[:40], c.get("roofline",{}).get("frac"), c.get("error"))
print(d.get("cpu_baseline"))
PY
cd /tmp
for wl in lap2d_10m lap3d_2m; do
  rocprofv3 --kernel-trace -d $O/prof_$wl -o p -- python $R/scripts/one_solve.py csr $wl > $O/run_$wl.log 2> $O/prof_$wl.log; tail -1 $O/run_$wl.log
  python $R/scripts/rocpd_summary.py $O/prof_$wl/p_results.db $O/kernel_stats_$wl.md > /dev/null; head -8 $O/kernel_stats_$wl.md; tail -1 $O/kernel_stats_$wl.md
  python $R/scripts/gap_analysis.py $O/prof_$wl/p_results.db $O/gap_analysis_$wl.md > /dev/null; tail -1 $O/gap_analysis_$wl.md
  rm -rf $O/prof_$wl
done
