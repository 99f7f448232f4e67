#!/bin/bash
# timing ablations of dcn_fwd4 (results wrong by construction): RVSR_DCN4_DBG bits 1 no weight DMA, 2 no x stores, 4 no x loads, 8 no barriers
for dbg in 0 1 2 4 8 3 7 15; do
  echo -n "dbg $dbg: "
  RVSR_DCN_FWD=4 RVSR_DCN4_NW=${NW:-8} RVSR_DCN4_DBG=$dbg timeout 120 python tools/dcn_micro.py --B 40 --iters 20 --fwd-only --ostd 1.25 2>&1 | tail -1
done
