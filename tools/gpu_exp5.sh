#!/bin/bash
OUT=gpurun_out/r01e; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
rocprofv3 -L > $OUT/counters.txt 2>&1
cd /tmp
i=0
for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM" \
           "SQ_WAIT_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VSKIPPED SQ_ITEMS SQ_WAVES_EQ_64" \
           "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCC_REQ_sum TCC_READ_sum TCC_BUSY_avr TCC_TAG_STALL_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_ACCESSES_sum"; do
  i=$((i+1))
  for fold in 4096 0; do
    timeout 200 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $R/$OUT/p${i}_f$fold -o p -- python $R/tools/spmm_sweep.py --only amazon-book --order degree --reps 3 --fold $fold > $R/$OUT/p${i}_f$fold.log 2>&1
    echo "== pass $i fold $fold exit $? : $pmc"
  done
done
