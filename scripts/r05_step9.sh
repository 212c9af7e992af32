#!/bin/bash
# round 5, step 9: what bounds the row-pattern product?  (a) the stencil probe: its ingredients one at a time; (b) counters of the kernel itself
R=$PWD; O=$R/gpurun_out/r05_step9; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 300 scripts/probes/stencil_probe 2>&1 | tee $O/stencil_probe.txt
cd /tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_TCC_WRITE_REQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $O/pmc$i -o p -- python $R/scripts/spmv_format_perf.py 20 > $O/pmc$i.log 2>&1
  DB=$(find $O/pmc$i -name "*.db" | head -1)
  if [ -n "$DB" ]; then python $R/scripts/pmc_summary.py $DB $O/pmc$i.md | grep "pat_kernel\|kernel |" | cut -c1-400; else echo "set $i ($set): no result"; tail -2 $O/pmc$i.log; fi
  rm -rf $O/pmc$i
done
