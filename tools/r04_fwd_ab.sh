#!/bin/bash
# A/B of the DCN forward generations inside one gpurun call (L1 shape, B = 40): RVSR_DCN_FWD=3 vs 4 (NW = 8 / 12 rows per workgroup, priority rotation on / off)
mkdir -p gpurun_out
for ostd in ${OSTD:-0.125 1.25}; do
  for cfg in "3 12 0" "4 8 0" "4 8 1" "4 12 0" "4 12 1"; do
    set -- $cfg
    echo -n "ostd $ostd gen $1 nw $2 prio $3: "
    RVSR_DCN_FWD=$1 RVSR_DCN4_NW=$2 RVSR_DCN4_PRIO=$3 timeout 120 python tools/dcn_micro.py --B 40 --iters 20 --fwd-only --ostd $ostd 2>&1 | tail -1
  done
done
