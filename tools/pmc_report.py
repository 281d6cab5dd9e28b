"""Per-kernel summary of rocprofv3 --pmc passes (counter_collection CSVs), and the bench's traffic file.

    python tools/pmc_report.py <dir-with-passes> [--kernels ft_gemm_rows_b3,ft_rnn_fwd_persist,...] [--min-grid N]
    python tools/pmc_report.py <dir> --traffic-json profiles/r02_pmc_bank_fwd.json --kernel ft_gemm_rows_b3 \
        --grid-wgs 3376 --shape 32,841,80,256,8

Each pass directory holds one `*counter_collection.csv` (one row per dispatch and counter).  Counters of different
passes are joined per kernel name (dispatch ids differ between passes, so the join is on (kernel, grid, nth launch)).
Units: FETCH_SIZE / WRITE_SIZE in KiB (rocprofv3); FETCH_SIZE is DOUBLED for the byte estimate, as
MI355X_MICROARCH.md (HBM section) prescribes for wide coalesced reads on gfx950; SQ_VALU_MFMA_BUSY_CYCLES counts cycles
summed over SIMDs, SQ_BUSY_CYCLES / SQ_WAVE_CYCLES quad-cycles (see the guide's cycle-constants table).
"""
import argparse
import collections
import csv
import glob
import json
import os
import re


def short(name):
    m = re.search(r'(ft_\w+)(<[^>(]*>)?', name)
    return (m.group(1) + (m.group(2) or '')) if m else name[:60]


def load(root):
    """{(kernel, grid): {counter: [values in launch order]}}"""
    data = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True)):
        rows = list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: int(r.get('Dispatch_Id', 0)))
        for r in rows:
            key = (short(r['Kernel_Name']), int(r['Grid_Size']) // max(int(r['Workgroup_Size']), 1))
            data[key][r['Counter_Name']].append(float(r['Counter_Value']))
    return data


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('root')
    ap.add_argument('--kernels', default='')
    ap.add_argument('--min-launch-share', type=float, default=0.0)
    ap.add_argument('--traffic-json')
    ap.add_argument('--kernel')
    ap.add_argument('--grid-wgs', type=int)
    ap.add_argument('--shape')
    ap.add_argument('--pick', default='0/1', help='i/n: of the matching launches take every n-th starting at i (a step may '
                    'hold several launches of one kernel and grid: the conv-bank forward is the first of two)')
    args = ap.parse_args()
    data = load(args.root)
    want = [k for k in args.kernels.split(',') if k]
    print(f'{"kernel":58s} {"wgs":>6s} {"n":>4s}  counters (mean per launch)')
    for (k, wgs), d in sorted(data.items()):
        if want and not any(w in k for w in want):
            continue
        n = max(len(v) for v in d.values())
        parts = []
        for c, v in sorted(d.items()):
            parts.append(f'{c}={sum(v) / len(v):.4g}')
        print(f'{k:58s} {wgs:6d} {n:4d}  ' + '  '.join(parts))
        if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
            fe, wr = sum(d['FETCH_SIZE']) / len(d['FETCH_SIZE']), sum(d['WRITE_SIZE']) / len(d['WRITE_SIZE'])
            print(f'{"":58s} {"":6s} {"":4s}  -> HBM-side bytes/launch = 2 x {fe * 1024 / 1e6:.1f} MB read + '
                  f'{wr * 1024 / 1e6:.1f} MB written = {(2 * fe + wr) * 1024 / 1e6:.1f} MB')
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and 'SQ_BUSY_CYCLES' in d:
            mb = sum(d['SQ_VALU_MFMA_BUSY_CYCLES']) / len(d['SQ_VALU_MFMA_BUSY_CYCLES'])
            sb = sum(d['SQ_BUSY_CYCLES']) / len(d['SQ_BUSY_CYCLES'])
            print(f'{"":58s} {"":6s} {"":4s}  -> MFMA busy / SQ busy = {mb / max(sb, 1):.3f} (raw ratio of the two counters)')
    if args.traffic_json:
        key = next(((k, w) for (k, w) in data if args.kernel in k and (args.grid_wgs is None or w == args.grid_wgs)
                    and 'FETCH_SIZE' in data[(k, w)] and 'WRITE_SIZE' in data[(k, w)]), None)
        if key is None:
            raise SystemExit('no kernel with both FETCH_SIZE and WRITE_SIZE matches')
        d = data[key]
        pi, pn = (int(v) for v in args.pick.split('/'))
        fsel, wsel = d['FETCH_SIZE'][pi::pn], d['WRITE_SIZE'][pi::pn]
        fe, wr = sum(fsel) / len(fsel), sum(wsel) / len(wsel)
        rec = {'kernel': key[0], 'grid_workgroups': key[1], 'shape': [int(v) for v in args.shape.split(',')],
               'launches': len(fsel), 'picked': args.pick, 'fetch_size_kib': fe, 'write_size_kib': wr,
               'traffic_bytes_per_launch': (2 * fe + wr) * 1024,
               'method': 'separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes; FETCH_SIZE x 2 (gfx950, wide '
                         'coalesced reads) + WRITE_SIZE, KiB -> bytes'}
        json.dump(rec, open(args.traffic_json, 'w'), indent=1)
        print('wrote', args.traffic_json, rec['traffic_bytes_per_launch'] / 1e6, 'MB/launch')


if __name__ == '__main__':
    main()
