"""Where is the GPU idle inside one train step?  Reads a rocprofv3 --kernel-trace CSV, takes the LAST step (between
the last two Adam launches) and prints per-stream busy time, the time no kernel at all was running, and the largest
gaps with the kernels on either side.   usage: python tools/trace_gaps.py <kernel_trace.csv> [n_gaps]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ng = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'ft_adam_kernel' in r['Kernel_Name']]
last = rows[idx[-2] + 1: idx[-1] + 1] if len(idx) >= 2 else rows


def short(n):
    m = re.search(r'(ft_\w+)(<[^>]*>)?', n)
    return (m.group(1) + (m.group(2) or '')) if m else n[:50]


t0 = int(last[0]['Start_Timestamp'])
span = int(last[-1]['End_Timestamp']) - t0
skey = 'Stream_Id' if 'Stream_Id' in last[0] else ('Queue_Id' if 'Queue_Id' in last[0] else None)
per = collections.defaultdict(int)
for r in last:
    per[r.get(skey, '?') if skey else '?'] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
print(f'last step: {len(last)} launches, span {span / 1e6:.2f} ms; columns: {list(last[0].keys())}')
for k, v in sorted(per.items(), key=lambda kv: -kv[1]):
    print(f'  {skey} {k}: busy {v / 1e6:.2f} ms')
# union of busy intervals
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in last)
gaps = []
cur_end, cur_name = ev[0][1], ev[0][2]
idle = 0
for s, e, n in ev[1:]:
    if s > cur_end:
        gaps.append((s - cur_end, cur_end - t0, cur_name, n))
        idle += s - cur_end
    if e > cur_end:
        cur_end, cur_name = e, n
print(f'no kernel running: {idle / 1e6:.2f} ms in {len(gaps)} gaps')
hist = collections.Counter()
for g in gaps:
    hist[min(g[0] // 2000 * 2, 20)] += g[0]
print('idle by gap size (us bucket -> total ms):', {f'{k}+': round(v / 1e6, 2) for k, v in sorted(hist.items())})
for g in sorted(gaps, reverse=True)[:ng]:
    print(f'  {g[0] / 1e3:8.1f} us at +{g[1] / 1e6:6.2f} ms   after {g[2]}   before {g[3]}')
