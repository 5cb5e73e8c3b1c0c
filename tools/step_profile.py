#!/usr/bin/env python3
"""Per-kernel breakdown of ONE training step from a rocprofv3 kernel trace (steps delimited by the optimizer kernel)."""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'multi_tensor' in r['Kernel_Name'] or 'k_adam' in r['Kernel_Name']]
groups = []
for i in idx:
    if not groups or i - groups[-1][-1] > 1: groups.append([i])
    else: groups[-1].append(i)
a, b = groups[-3][-1] + 1, groups[-2][-1] + 1
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp']); t1 = int(step[-1]['End_Timestamp'])
print('one train step: kernels', len(step), 'span ms %.3f' % ((t1 - t0) / 1e6),
      'busy ms %.3f' % (sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step) / 1e6))
agg = collections.defaultdict(list)
for r in step:
    agg[(r['Kernel_Name'][:60], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:n]:
    print(k, len(v), 'avg_us %.1f' % (sum(v) / len(v) / 1e3), 'total_ms %.3f' % (sum(v) / 1e6))
