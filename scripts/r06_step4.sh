#!/bin/bash
# Round 6, step 4: grid sizes of the TN panel kernels (complex vector-unit kernel at 1 / 2 / 3 workgroups per CU; the real
# matrix-core kernel at 1 / 2 / 3 / 4 and the real vector-unit kernels), the scatter probe of configs[4], configs[3] timing.
R=$PWD; O=$R/gpurun_out; TAG=r06
( for b in 1 2 3; do HIPK_ZDOTS_BPC=$b python scripts/zpanel_perf.py 2>&1 | grep complex; done
  for b in 1 2 3 4; do HIPK_MFMA_BPC=$b python scripts/zpanel_perf.py 2>&1 | grep real; done
  HIPK_NO_MFMA=1 python scripts/zpanel_perf.py 2>&1 | grep real ) > $O/${TAG}_zpanel_perf2.txt 2>&1
cat $O/${TAG}_zpanel_perf2.txt
( scripts/probes/scatter_probe 2000000 8000000; scripts/probes/scatter_probe 8000000 2000000 ) > $O/${TAG}_scatter_probe.txt 2>&1
cat $O/${TAG}_scatter_probe.txt
python scripts/config4_run.py 2>&1 | cut -c1-200 > $O/${TAG}_config4_bpc2.log; cat $O/${TAG}_config4_bpc2.log
