#!/bin/bash
# DCN pack backward at the L1 shape (B = 40), i.i.d. offsets: device-selected window against forced windows; per-kernel stats of two cases
mkdir -p gpurun_out
for ostd in 1.25; do
  for h in auto 4; do
    echo -n "ostd $ostd halo $h fwd+bwd: "
    if [ $h = auto ]; then timeout 120 python tools/dcn_micro.py --B 40 --iters 10 --ostd $ostd 2>&1 | tail -1
    else RVSR_DCN5_HALO=$h timeout 120 python tools/dcn_micro.py --B 40 --iters 10 --ostd $ostd 2>&1 | tail -1; fi
  done
done
cd /tmp && export TMPDIR=/tmp
for sm in 0 16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/bwdprof_sm$sm -o p -- python $GRAFT_REPO_ROOT/tools/dcn_micro.py --B 40 --iters 10 --ostd 1.25 --smooth $sm > /dev/null 2>&1
  echo "== smooth $sm"; python - <<PY
import csv,glob
f=glob.glob('$GRAFT_REPO_ROOT/gpurun_out/bwdprof_sm$sm/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r['Name'][:70], r['Calls'], '%.3f ms avg'%(float(r['AverageNs'])/1e6))
PY
done
