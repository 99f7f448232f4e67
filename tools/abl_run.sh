#!/bin/bash
# time the DCN forward micro-benchmark with each ablation build (tools/build_variant.sh abl<n> -DRVSR_ABL=<n>)
for v in hip "$@"; do
  echo -n "$v: "; RVSR_SO=$PWD/realvsr_amd/csrc/librealvsr_$v.so python tools/dcn_micro.py --fwd-only --B 40 --ostd 0.1 --iters 20 2>&1 | tail -1
done
