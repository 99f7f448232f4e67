#!/bin/bash
# A/B of the DCN input-gradient kernels inside one gpurun call: per-kernel averages (rocprofv3 --kernel-trace --stats) of
# tools/dcn_micro.py at several offset scales.   usage: tools/r03_bwd_ab.sh "<gens: 5 6>" "<ostd list>" [micro flags]
GENS="$1"; OSTDS="$2"; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
mkdir -p $R/gpurun_out/bwdab
for o in $OSTDS; do
  for g in $GENS; do
    out=$R/gpurun_out/bwdab/g${g}_o${o}
    rm -rf $out
    RVSR_DCN_BWD=$g timeout 300 rocprofv3 --kernel-trace --stats -d $out -o t --output-format csv -- python $R/tools/dcn_micro.py --ostd $o --iters 3 "$@" > $out.log 2>&1
    f=$(find $out -name "*kernel_stats.csv" | head -1)
    echo "== gen $g ostd $o: $(grep 'dcn pack' $out.log)"
    python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r['Name']
    if 'dcn_' in n or 'pack_weights' in n or 'probe' in n:
        print('   %-60s calls %4s  avg %10.1f us  total %10.1f us' % (n[:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3))
PY
  done
done
