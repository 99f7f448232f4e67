#!/bin/bash
# round 5: compile-time ablations of dcn_bwdin6 (tools/build_variant6.sh b<bits> -DRVSR_ABL6=<bits>), per-kernel averages under rocprofv3
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
L=gpurun_out/r05_abl6.log
: > $L
for ostd in 1.25 0.125; do
for v in ${VARIANTS:-hip b1}; do
  rm -rf gpurun_out/prof_abl6
  RVSR_SO=$PWD/realvsr_amd/csrc/librealvsr_$v.so timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_abl6 --output-format csv -- python tools/dcn_micro.py --B 40 --iters 6 --ostd $ostd > /dev/null 2>&1
  f=$(ls gpurun_out/prof_abl6/*/*kernel_stats.csv 2>/dev/null | head -1)
  python - "$f" "$ostd" "$v" >> $L <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'dcn_bwdin6' in r['Name'] or 'dcn_bwdw' in r['Name']]
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
print('ostd %s [%s] ' % (sys.argv[2], sys.argv[3]) + '; '.join('%s avg %.3f ms' % (r['Name'].split('(')[0].replace('void ', ''), float(r['AverageNs']) / 1e6) for r in rows[:2]))
PY
done
done
rm -rf gpurun_out/prof_abl6
cat $L
