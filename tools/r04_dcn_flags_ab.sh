for v in hip dilp hip dilp; do
  echo -n "$v fwd 0.125: "; RVSR_SO=$PWD/realvsr_amd/csrc/librealvsr_$v.so timeout 120 python tools/dcn_micro.py --B 40 --iters 20 --fwd-only --ostd 0.125 2>&1 | tail -1
  echo -n "$v fwd 1.25: "; RVSR_SO=$PWD/realvsr_amd/csrc/librealvsr_$v.so timeout 120 python tools/dcn_micro.py --B 40 --iters 20 --fwd-only --ostd 1.25 2>&1 | tail -1
  echo -n "$v fwd+bwd 1.25: "; RVSR_SO=$PWD/realvsr_amd/csrc/librealvsr_$v.so timeout 120 python tools/dcn_micro.py --B 40 --iters 40 --ostd 1.25 2>&1 | tail -1
done
