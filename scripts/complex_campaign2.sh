#!/bin/bash
# Round 3, second GPU pass on the complex path: where the native configs[3] solve spends its wall-clock, the crash
# of the grouped test run, the fixture tests.
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
L=$O/r03_complex_campaign2.log
: > $L
echo "== grouped run that dumped core in the first pass (full output)" >> $L
timeout 900 python -X faulthandler -m pytest tests/test_complex_gpu.py tests/test_c_examples_gpu.py tests/test_kernels_gpu.py -q -p no:cacheprovider -x > $O/r03_grouped_run.log 2>&1; echo "rc=$?" >> $O/r03_grouped_run.log
head -60 $O/r03_grouped_run.log >> $L; tail -5 $O/r03_grouped_run.log >> $L
echo "== configs[3] native, host timing" >> $L
HIPK_HOST_TIMING=1 FORM=native timeout 300 python scripts/config4_run.py 2>&1 | cut -c1-300 >> $L
echo "== rocprof gap analysis of the native run" >> $L
FORM=native timeout 300 rocprofv3 --kernel-trace -d $O/r03_prof_c4 -o c4 -- python scripts/config4_run.py > $O/r03_config4_under_rocprof.log 2>&1
python scripts/gap_analysis.py $O/r03_prof_c4/c4_results.db $O/r03_config4_native_gaps.md --max-gap-us 100000 > /dev/null 2>&1; cat $O/r03_config4_native_gaps.md >> $L
python - >> $L 2>&1 <<'PY'
import sqlite3
db = sqlite3.connect("gpurun_out/r03_prof_c4/c4_results.db")
rows = db.execute("select name, start, end from kernels order by start").fetchall()
import re
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "")[:40]
big = sorted(((rows[i + 1][1] - rows[i][2]) / 1e3, i) for i in range(len(rows) - 1))[-40:]
print("largest idle gaps (us): previous -> next")
for g, i in sorted(big, key=lambda t: t[1]):
    print(f"{g:10.1f}  {short(rows[i][0])} -> {short(rows[i+1][0])}")
PY
rm -rf $O/r03_prof_c4
cat $L
