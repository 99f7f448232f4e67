#!/bin/bash
# timing ablations of dcn_fwd3 (scratch builds librealvsr_a3_<bits>.so, tools/build_variant.sh; results wrong by construction)
for ostd in 0.125 1.25; do
  echo -n "ostd $ostd fwd3: "; RVSR_DCN_FWD=3 timeout 120 python tools/dcn_micro.py --B 40 --iters 20 --fwd-only --ostd $ostd 2>&1 | tail -1
  for b in "$@"; do
    echo -n "ostd $ostd fwd3 abl $b: "; RVSR_SO=$PWD/realvsr_amd/csrc/librealvsr_a3_$b.so RVSR_DCN_FWD=3 timeout 120 python tools/dcn_micro.py --B 40 --iters 20 --fwd-only --ostd $ostd 2>&1 | tail -1
  done
done
