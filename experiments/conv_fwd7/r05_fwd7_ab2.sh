#!/bin/bash
mkdir -p gpurun_out
{
python tools/r05_fwd7_dbg.py 2>&1 | grep differing
for r in 1 2; do
for v in fwd5 hip prio7; do
  so=$PWD/realvsr_amd/csrc/librealvsr_$v.so; sw=1
  if [ $v = fwd5 ]; then so=$PWD/realvsr_amd/csrc/librealvsr_hip.so; sw=0; fi
  echo -n "$v fwd: "; RVSR_SO=$so RVSR_CONV_FWD7=$sw timeout 120 python tools/conv_micro.py --iters 30 2>&1 | tail -1
done
done
RVSR_SO=$PWD/realvsr_amd/csrc/librealvsr_tl7.so python tools/conv7_timeline.py 2>&1 | tail -50
} > gpurun_out/r05_fwd7_ab2.log 2>&1
cat gpurun_out/r05_fwd7_ab2.log
