#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_step5; mkdir -p $O
export PRIMME_AMD_PRELAUNCH_STATS=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "rr_arrow" > $O/tests_k.txt 2>&1; echo "rr kernel test exit $?"; tail -3 $O/tests_k.txt
for nb in 0 1 0 1; do
  if [ $nb = 1 ]; then export PRIMME_AMD_NO_PRELAUNCH=1; else unset PRIMME_AMD_NO_PRELAUNCH; fi
  timeout 600 python bench.py --workload lap3d_2m --steps 3 --warmup 1 --no-extra-configs --no-cpu-baseline > $O/b.json 2> $O/b.err
  echo "configs1 NO_PRELAUNCH=$nb: $(python -c "import json;d=json.load(open('$O/b.json'));print(d['value'], d['ms_per_step'], d['config']['us_per_outer_iteration'], d['config']['outer_iterations'])")"; grep "enqueued ahead" $O/b.err | tail -1
done
unset PRIMME_AMD_NO_PRELAUNCH
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $R/$O/trace -o c1 -- python $R/scripts/one_solve.py csr lap3d_2m > $R/$O/trace.log 2>&1; echo "trace exit $?"; tail -2 $R/$O/trace.log
cd $R
DB=$(find $O/trace -name "*.db" | head -1)
python scripts/rocpd_summary.py $DB $O/c1_kernel_stats.md | head -16
python scripts/rocpd_gaps.py $DB $O/c1_gaps.md | head -30
rm -rf $O/trace
