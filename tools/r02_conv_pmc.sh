#!/bin/bash
O=gpurun_out/conv_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"
P3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES"
i=0
for P in "$P1" "$P2" "$P3" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); d=$O/tmp
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $d -- python tools/conv_micro.py --bwd > /dev/null 2>&1
  f=$(find $d -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py "$f" conv > $O/p$i.txt && cat $O/p$i.txt | grep -v pack
  rm -rf $d
done
