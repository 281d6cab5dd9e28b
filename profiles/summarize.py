"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel time of the LAST train step (between the last two
Adam launches).  usage: python profiles/summarize.py <kernel_trace.csv> [top_n]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'ft_adam_kernel' in r['Kernel_Name']]
last = rows[idx[-2] + 1: idx[-1] + 1] if len(idx) >= 2 else rows
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in last)
span = int(last[-1]['End_Timestamp']) - int(last[0]['Start_Timestamp'])
print(f'last step: {len(last)} launches, busy {busy / 1e6:.2f} ms, span {span / 1e6:.2f} ms')
agg = collections.defaultdict(lambda: [0, 0])
for r in last:
    m = re.search(r'(ft_\w+)(<[^>]*>)?', r['Kernel_Name'])
    name = (m.group(1) + (m.group(2) or '')) if m else r['Kernel_Name'][:60]
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    agg[name][0] += d
    agg[name][1] += 1
print(f'{"total_us":>10} {"calls":>6} {"avg_us":>9}  kernel')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f'{v[0] / 1e3:10.1f} {v[1]:6d} {v[0] / 1e3 / v[1]:9.2f}  {k}')
