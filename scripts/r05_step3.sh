#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_step3; mkdir -p $O
export PRIMME_AMD_TEST_NORMAL_EXIT=1
timeout 900 python -m pytest tests/test_reference_kernels.py tests/test_comm_gpu.py tests/test_solver_gpu.py -m gpu -q -p no:cacheprovider > $O/tests_a.txt 2>&1; echo "tests_a exit $?"; tail -5 $O/tests_a.txt
cat gpurun_out/exact_history_gpu.json
timeout 600 python -m pytest tests/test_multirank_ipc_gpu.py -m gpu -q -p no:cacheprovider -k "collectives or wraps" > $O/tests_b.txt 2>&1; echo "tests_b exit $?"; tail -5 $O/tests_b.txt
for nx in 2000000 8000000; do timeout 120 scripts/probes/gather_probe $nx 2>&1 | tee -a $O/gather_probe.txt; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OLDPWD/$O/pmc_gather -o gp -- $OLDPWD/scripts/probes/gather_probe 2000000 > $OLDPWD/$O/pmc_gather.log 2>&1; echo "pmc exit $?"
cd $OLDPWD
DB=$(find $O/pmc_gather -name "*.db" | head -1); echo "db: $DB"
[ -n "$DB" ] && python scripts/pmc_summary.py $DB $O/pmc_gather_fetch.md
export PRIMME_AMD_LIB=$PWD/primme_amd/variants/libprimme_amd_pat_ntst.so
echo "=== pat_ntst" | tee -a $O/spmv_format_perf.txt
timeout 300 python scripts/spmv_format_perf.py 2>&1 | grep -v amdgpu.ids | tee -a $O/spmv_format_perf.txt | grep -E "format 2|identical"
rm -rf $O/pmc_gather/*/*.csv 2>/dev/null; du -sh $O
