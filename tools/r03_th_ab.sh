#!/bin/bash
# A/B of an environment switch on the DCN micro-benchmark: per-kernel rocprofv3 averages.  tools/r03_th_ab.sh "<ENV_A>" "<ENV_B>" "<ostd list>" [micro flags]
A="$1"; B="$2"; OSTDS="$3"; shift 3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
mkdir -p $R/gpurun_out/thab
for o in $OSTDS; do
  for v in A B; do
    [ $v = A ] && E="$A" || E="$B"
    out=$R/gpurun_out/thab/${v}_o${o}; rm -rf $out
    env $E timeout 300 rocprofv3 --kernel-trace --stats -d $out -o t --output-format csv -- python $R/tools/dcn_micro.py --ostd $o --iters 3 "$@" > $out.log 2>&1
    f=$(find $out -name "*kernel_stats.csv" | head -1)
    python - "$f" "$v [$E] ostd $o" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if ('dcn_bwdin' in r['Name'] or 'dcn_bwdw' in r['Name'] or 'dcn_fwd' in r['Name']) and float(r['AverageNs']) > 50e3:
        print('%-28s %-44s avg %9.1f us' % (sys.argv[2], r['Name'].split('(')[0][-44:], float(r['AverageNs']) / 1e3))
PY
  done
done
