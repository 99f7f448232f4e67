#!/usr/bin/env python
"""Sum rocprofv3 counter_collection.csv per (kernel, counter): mean per dispatch over the dcn_* kernels."""
import csv
import collections
import sys

acc = collections.defaultdict(lambda: [0.0, 0])
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        name = row.get('Kernel_Name', '')
        if 'dcn' not in name and (len(sys.argv) < 3 or sys.argv[2] not in name):
            continue
        key = (name.split('(')[0][:60], row['Counter_Name'])
        acc[key][0] += float(row['Counter_Value'])
        acc[key][1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print('%-62s %-28s %16.0f  (mean of %d dispatches)' % (k, c, v / n, n))
