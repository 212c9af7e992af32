export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -p no:cacheprovider -k "ritz" > $O/r02_t13.log 2>&1; tail -3 $O/r02_t13.log
python scripts/kernel_perf.py 2>&1 | grep -i "restart\|ritz RES k=15\|overlaps+W'r k=15"
