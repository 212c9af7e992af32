#!/bin/bash
# configs[3] native: non-temporal panel loads in the complex kernels (default) against the plain-load build
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp; L=$O/r03_znt_ab.log; : > $L
timeout 300 python -X faulthandler -m pytest tests/test_kernels_complex_gpu.py -q -p no:cacheprovider 2>&1 | tail -1 >> $L
echo "== default (non-temporal)" >> $L
FORM=native timeout 200 python scripts/config4_run.py 2>&1 | grep rep | cut -c1-70 >> $L
echo "== plain loads" >> $L
PRIMME_AMD_LIB=$PWD/primme_amd/variants/libprimme_amd_znt0.so FORM=native timeout 200 python scripts/config4_run.py 2>&1 | grep rep | cut -c1-70 >> $L
echo "== default (non-temporal)" >> $L
FORM=native timeout 200 python scripts/config4_run.py 2>&1 | grep rep | cut -c1-70 >> $L
cat $L
