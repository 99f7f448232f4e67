#!/bin/bash
# time the 3x3 conv micro-benchmark (tools/conv_micro.py) with the product library and each named variant build
for v in hip "$@"; do
  echo -n "$v: "; RVSR_SO=$PWD/realvsr_amd/csrc/librealvsr_$v.so python tools/conv_micro.py --iters 20 2>&1 | tail -1
done
