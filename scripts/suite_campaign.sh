#!/bin/bash
# full GPU suite N times + the grouped run, full logs kept (gpurun; arg 1 = tag, arg 2 = repetitions)
TAG=${1:-r03_suite}; N=${2:-2}
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
L=$O/${TAG}.log; : > $L
for i in $(seq 1 $N); do
  timeout 900 python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider > $O/${TAG}_full$i.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_full$i.log
  echo "== full suite $i" >> $L; tail -4 $O/${TAG}_full$i.log | cut -c1-200 >> $L
  timeout 300 python -X faulthandler -m pytest tests/test_complex_gpu.py tests/test_c_examples_gpu.py tests/test_kernels_gpu.py -q -p no:cacheprovider > $O/${TAG}_group$i.log 2>&1; echo "rc=$?" >> $O/${TAG}_group$i.log
  echo "== grouped $i" >> $L; tail -3 $O/${TAG}_group$i.log | cut -c1-200 >> $L
done
cat $L
