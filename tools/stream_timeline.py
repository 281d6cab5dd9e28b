"""Timeline of ONE stream inside the last train step of a rocprofv3 --kernel-trace CSV: every launch of the busiest
stream (or the one given) in start order with its offset, duration and the idle time before it, plus what the OTHER
streams were running during each idle interval -- shows where the critical stream waits and for what.
usage: python tools/stream_timeline.py <kernel_trace.csv> [stream_id] [min_gap_us]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'ft_adam_kernel' in r['Kernel_Name']]
last = rows[idx[-2] + 1: idx[-1] + 1] if len(idx) >= 2 else rows
skey = 'Stream_Id' if 'Stream_Id' in last[0] else 'Queue_Id'


def short(n):
    m = re.search(r'(ft_\w+)', n)
    return m.group(1).replace('_kernel', '') if m else n[:28]


busy = collections.defaultdict(int)
for r in last:
    busy[r[skey]] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
sid = sys.argv[2] if len(sys.argv) > 2 else max(busy, key=busy.get)
min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 15.0
mine = [r for r in last if r[skey] == sid]
# drop the host pause some traces contain (GC freeze, probes): restart the clock after any gap > 5 ms
t0 = int(mine[0]['Start_Timestamp'])
others = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r[skey]) for r in last if r[skey] != sid]
print(f'stream {sid}: {len(mine)} launches, busy {busy[sid] / 1e6:.2f} ms; other streams: '
      + ', '.join(f'{k}: {v / 1e6:.2f} ms' for k, v in busy.items() if k != sid))
prev_end = t0
idle_total = 0
run_name, run_n, run_us, run_start = None, 0, 0.0, 0
out = []


def flush():
    if run_name is not None:
        out.append(f'  +{run_start / 1e6:7.3f} ms  {run_us:8.1f} us  {run_name}' + (f' x{run_n}' if run_n > 1 else ''))


for r in mine:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3
    name = short(r['Kernel_Name'])
    if gap >= min_gap:
        flush()
        run_name = None
        if gap > 5000:
            out.append(f'  ---- host pause {gap / 1e3:.1f} ms ----')
        else:
            idle_total += gap
            ov = collections.defaultdict(float)
            for (os_, oe, on, osid) in others:
                lo, hi = max(os_, prev_end), min(oe, s)
                if hi > lo:
                    ov[f'{on}@{osid}'] += (hi - lo) / 1e3
            what = ', '.join(f'{k} {v:.0f}' for k, v in sorted(ov.items(), key=lambda kv: -kv[1])[:3]) or 'nothing anywhere'
            out.append(f'  .... idle {gap:7.1f} us   (meanwhile: {what})')
    elif gap > 0:
        idle_total += gap
    if name == run_name:
        run_n += 1
        run_us += (e - s) / 1e3
    else:
        flush()
        run_name, run_n, run_us, run_start = name, 1, (e - s) / 1e3, s - t0
    prev_end = max(prev_end, e)
flush()
print('\n'.join(out))
print(f'idle on this stream (gaps < 5 ms): {idle_total / 1e3:.2f} ms')
