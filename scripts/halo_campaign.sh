#!/bin/bash
# Round-3 hunt for the intermittent failure of test_csr_row_slabs_with_halo (run through gpurun).
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
L=$O/r03_halo_campaign.log
: > $L
echo "== in-process loops, legacy NULL-stream uploads" >> $L
HIPK_LEGACY_UPLOAD=1 timeout 300 python scripts/halo_repro.py --iters 400 >> $L 2>&1
HIPK_LEGACY_UPLOAD=1 timeout 300 python scripts/halo_repro.py --iters 200 --dtype f32 >> $L 2>&1
echo "== in-process loops, uploads on the context stream" >> $L
timeout 300 python scripts/halo_repro.py --iters 400 >> $L 2>&1
timeout 300 python scripts/halo_repro.py --iters 200 --dtype f32 >> $L 2>&1
echo "== fresh processes: pytest test_kernels_gpu.py (whole file, legacy uploads), 6 runs" >> $L
for i in 1 2 3 4 5 6; do
  HIPK_LEGACY_UPLOAD=1 timeout 300 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider 2>&1 | tail -3 >> $L
done
echo "== fresh processes: pytest test_kernels_gpu.py (whole file, context-stream uploads), 3 runs" >> $L
for i in 1 2 3; do
  timeout 300 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider 2>&1 | tail -3 >> $L
done
echo "== full GPU suite, context-stream uploads" >> $L
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r03_gpu_suite_run1.log 2>&1; echo "pytest rc=$?" >> $O/r03_gpu_suite_run1.log
tail -5 $O/r03_gpu_suite_run1.log >> $L
cat $L
