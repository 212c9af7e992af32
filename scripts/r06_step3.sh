#!/bin/bash
# Round 6, step 3 (through gpurun): the complex TN panel product in isolation (vector units against matrix cores, grid
# sizes), configs[1] with / without the quiet second stage and the one-launch tail, the reference-indexing knob on the device.
R=$PWD; O=$R/gpurun_out; TAG=r06
python -m pytest tests/test_solver_gpu.py -q -m gpu -x -k "own_indexing" -rP 2>&1 | grep -E "passed|failed|indexing on the device|blk8" > $O/${TAG}_s3_tests.log
python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "tail_without or rr_arrow" 2>&1 | tail -3 >> $O/${TAG}_s3_tests.log
cat $O/${TAG}_s3_tests.log
( python scripts/zpanel_perf.py
  HIPK_ZDOTS_BPC=2 python scripts/zpanel_perf.py 2>&1 | grep complex
  HIPK_ZDOTS_BPC=8 python scripts/zpanel_perf.py 2>&1 | grep complex
  HIPK_ZMFMA=1 python scripts/zpanel_perf.py 2>&1 | grep complex
  HIPK_ZMFMA=1 HIPK_ZMFMA_BPC=3 python scripts/zpanel_perf.py 2>&1 | grep complex
  HIPK_ZMFMA=1 HIPK_ZMFMA_BPC=4 python scripts/zpanel_perf.py 2>&1 | grep complex ) > $O/${TAG}_zpanel_perf.txt 2>&1
cat $O/${TAG}_zpanel_perf.txt
( echo "configs[1], 5 solves per process, solver seconds; default:"; REPS=5 python scripts/one_solve.py csr lap3d_2m
  echo "PRIMME_AMD_LOUD_FIN=1:"; PRIMME_AMD_LOUD_FIN=1 REPS=5 python scripts/one_solve.py csr lap3d_2m
  echo "HIPK_NO_TAIL_DEFER=1:"; HIPK_NO_TAIL_DEFER=1 REPS=5 python scripts/one_solve.py csr lap3d_2m
  echo "default again:"; REPS=5 python scripts/one_solve.py csr lap3d_2m ) > $O/${TAG}_configs1_tail_ab.txt 2>&1
cat $O/${TAG}_configs1_tail_ab.txt
