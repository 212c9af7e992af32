#!/bin/bash
# Round 6, step 5: the whole GPU suite on the current tree, configs[3] with 2 / 4 workgroups per CU in the complex TN kernel
# (same box, alternating), configs[2] with 2 / 3 workgroups per CU in the real matrix-core kernel, then the bench line.
R=$PWD; O=$R/gpurun_out; TAG=r06
python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > $O/${TAG}_gpu_suite.log; cat $O/${TAG}_gpu_suite.log
( for i in 1 2; do for b in 2 4; do echo "HIPK_ZDOTS_BPC=$b"; HIPK_ZDOTS_BPC=$b python scripts/config4_run.py 2>&1 | grep -o "'seconds': [0-9.]*\|'outer': [0-9]*" | tr '\n' ' '; echo; done; done ) > $O/${TAG}_config4_bpc_ab.txt 2>&1; cat $O/${TAG}_config4_bpc_ab.txt
( for i in 1 2; do for b in 3 2; do echo "HIPK_MFMA_BPC=$b"; HIPK_MFMA_BPC=$b python scripts/config3_run.py --reps 3 2>&1 | tail -1 | cut -c1-120; done; done ) > $O/${TAG}_config3_mfma_bpc_ab.txt 2>&1; cat $O/${TAG}_config3_mfma_bpc_ab.txt
python bench.py > $O/${TAG}_bench2.json 2> $O/${TAG}_bench2.err; tail -c 1500 $O/${TAG}_bench2.json
