#!/bin/bash
# round 5: dcn_bwdw6 against dcn_bwdw4 (RVSR_DCN_BWDW=4): parity tests, then per-kernel averages under rocprofv3
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
L=gpurun_out/r05_bwdw6.log
: > $L
timeout 900 python -m pytest tests/test_gpu_dcn.py tests/test_gpu_dcn_fullsize.py -x -q -m gpu 2>&1 | tail -12 >> $L
for genw in 4 6; do
for ostd in 1.25 0.125; do
  rm -rf gpurun_out/prof_w6
  RVSR_DCN_BWDW=$genw timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_w6 --output-format csv -- python tools/dcn_micro.py --B 40 --iters 6 --ostd $ostd > /dev/null 2>&1
  f=$(ls gpurun_out/prof_w6/*/*kernel_stats.csv 2>/dev/null | head -1)
  python - "$f" "$ostd" "$genw" >> $L <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'dcn_bw' in r['Name'] or 'reduce_partials' in r['Name']]
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
print('ostd %s [RVSR_DCN_BWDW=%s] ' % (sys.argv[2], sys.argv[3]) + '; '.join('%s avg %.3f ms' % (r['Name'].split('(')[0].replace('void ', ''), float(r['AverageNs']) / 1e6) for r in rows[:4]))
PY
done
done
rm -rf gpurun_out/prof_w6
cat $L
