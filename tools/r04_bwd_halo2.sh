#!/bin/bash
for it in 10 40; do
  echo -n "auto iid iters $it: "; timeout 120 python tools/dcn_micro.py --B 40 --iters $it --ostd 1.25 2>&1 | tail -1
  echo -n "forced4 iid iters $it: "; RVSR_DCN5_HALO=4 timeout 120 python tools/dcn_micro.py --B 40 --iters $it --ostd 1.25 2>&1 | tail -1
  echo -n "auto smooth16 iters $it: "; timeout 120 python tools/dcn_micro.py --B 40 --iters $it --ostd 1.25 --smooth 16 2>&1 | tail -1
  echo -n "auto iid R4=0 iters $it: "; RVSR_DCN5_R4=0 timeout 120 python tools/dcn_micro.py --B 40 --iters $it --ostd 1.25 2>&1 | tail -1
done
