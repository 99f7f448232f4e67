#!/bin/bash
# quick kernel-stats of the default bench: tools/r02_stats.sh [extra bench flags]
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/stats; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $O/bench.json 2>/dev/null
cp "$(find $O/p -name '*kernel_stats.csv' | head -1)" $O/kernel_stats.csv
cp "$(find $O/p -name '*kernel_trace.csv' | head -1)" $O/kernel_trace.csv 2>/dev/null
rm -rf $O/p
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/stats/kernel_stats.csv')))
for r in rows[:32]:
    print('%-78s %5s %8.2f ms  avg %8.1f us' % (r['Name'][:78], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
# the torch elementwise adds, by duration
tr=list(csv.DictReader(open('gpurun_out/stats/kernel_trace.csv')))
adds=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in tr if 'CUDAFunctor_add' in r['Kernel_Name']]
adds.sort(reverse=True)
print('adds: n=%d total %.2f ms; top: %s' % (len(adds), sum(adds)/1e3, [round(a) for a in adds[:40]]))
cp=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in tr if 'elementwise_kernel_manual_unroll' in r['Kernel_Name'] or 'direct_copy' in r['Kernel_Name']]
cp.sort(reverse=True)
print('copies: n=%d total %.2f ms; top: %s' % (len(cp), sum(cp)/1e3, [round(a) for a in cp[:30]]))
PY
