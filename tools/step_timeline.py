#!/usr/bin/env python3
"""Timeline of ONE training step from a rocprofv3 kernel trace: start offset, duration, queue and name of every kernel
(steps delimited by the optimizer kernel).  usage: step_timeline.py <trace dir> [first] [count]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_adam' in r['Kernel_Name']]
a, b = idx[-3] + 1, idx[-2] + 1
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp'])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else len(step)
queues = {}
for i, r in enumerate(step[first:first + count]):
    q = queues.setdefault(r['Queue_Id'], len(queues))
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    name = r['Kernel_Name'].replace('mpnhip::', '').replace('void ', '')[:56]
    print(f"{first + i:4d} q{q} {s / 1e3:9.1f} {(e - s) / 1e3:7.1f}  {name}  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}")
