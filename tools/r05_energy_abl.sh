#!/bin/bash
# energy ablations of conv_fwd5 (scratch builds, results wrong by construction): package power / clock / launch rate in a 4 s loop + the 200-iteration micro
for v in hip "$@"; do echo "== $v"; RVSR_SO=$PWD/realvsr_amd/csrc/librealvsr_$v.so python tools/power_sample.py 2>&1 | grep "conv_fwd5 forward"; RVSR_SO=$PWD/realvsr_amd/csrc/librealvsr_$v.so python tools/conv_micro.py --iters 200 | tail -1; done
