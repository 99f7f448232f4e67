#!/bin/bash
# time the conv micro-benchmark (40 x 64 x 180 x 320, fwd) with the product library and each named variant build, twice
for rep in 1 2; do for v in hip "$@"; do
  echo -n "$v: "; RVSR_SO=$PWD/realvsr_amd/csrc/librealvsr_$v.so python tools/conv_micro.py --iters 20 2>&1 | tail -1
done; done
