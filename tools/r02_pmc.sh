#!/bin/bash
# SQ / TCC counter passes for the DCN kernels (tools/dcn_micro.py, L1 shape of config 2: B=40, C=64, 180x320)
# usage: tools/r02_pmc.sh <outdir> [ostd ...]
out=${1:-gpurun_out/r02_pmc}; shift
stds=${@:-0.1 1.25}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TCC|TCP|GRBM)_[A-Z0-9_]+" | sort -u > $out/counters_available.txt
wc -l $out/counters_available.txt
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"
P3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT"
P4="FETCH_SIZE"
P5="WRITE_SIZE"
for s in $stds; do
  i=0
  for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
    i=$((i+1))
    d=$out/std${s}_p$i
    timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $d -- python tools/dcn_micro.py --iters 2 --B 40 --ostd $s > $d.log 2>&1 || echo "pass $i std $s failed: $(tail -2 $d.log)"
    f=$(find $d -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python tools/pmc_summary.py "$f" > $out/std${s}_p$i.txt && cat $out/std${s}_p$i.txt
    rm -rf $d
  done
done
