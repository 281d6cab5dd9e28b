"""Per-launch report of the f32-MFMA GEMM family inside one train step (profiling aid).

    FT_GEMM_LOG=1 python tools/gemm_report.py run [--model forward|fastpitch] 2> gemm.log
    python tools/gemm_report.py summarize gemm.log

`run` executes warm-up steps, then ONE logged step between FTGEMM-BEGIN / FTGEMM-END markers (every GEMM launch is
timed with HIP events and synchronised, so the step itself is slow -- only the per-launch numbers mean anything).
"""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run(model_name):
    import torch
    from forwardtacotron_amd import data
    from forwardtacotron_amd.trainer import TrainStep
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    if model_name == 'fastpitch':
        from forwardtacotron_amd.fastpitch import FastPitch
        model = FastPitch(**data.FASTPITCH_MODEL).to(dev)
    else:
        from forwardtacotron_amd.model import ForwardTacotron
        model = ForwardTacotron(**data.SINGLESPEAKER_MODEL).to(dev)
    ts = TrainStep(model, lr=5e-5, train_cfg=dict(data.SINGLESPEAKER_TRAIN))
    batch = data.to_device(data.synthetic_batch(B=32, Tmax=128, n_mels=80, seed=0), dev)
    dur0 = batch['dur'].clone()
    for i in range(3):
        if i == 2:
            torch.cuda.synchronize()
            sys.stderr.write('FTGEMM-BEGIN\n')
            sys.stderr.flush()
        batch['dur'].copy_(dur0)
        ts.step(batch)
    torch.cuda.synchronize()
    sys.stderr.write('FTGEMM-END\n')
    sys.stderr.flush()


def summarize(path):
    rows = []
    on = False
    pat = re.compile(r'FTGEMM (\S+) M=(\d+) N=(\d+) K=(\d+) taps=(\d+) inst=(\d+) (\S+) us=([\d.]+) TF=([\d.]+)')
    for line in open(path, errors='replace'):
        if line.startswith('FTGEMM-BEGIN'):
            on, rows = True, []
        elif line.startswith('FTGEMM-END'):
            on = False
        elif on:
            m = pat.match(line)
            if m:
                rows.append((m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)),
                             int(m.group(6)), m.group(7), float(m.group(8)), float(m.group(9))))
    agg = collections.OrderedDict()
    for r in rows:
        key = r[:7]
        a = agg.setdefault(key, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += r[7]
        a[2] += r[7] * r[8]          # us * TF = MFLOP-ish weight
    tot = sum(a[1] for a in agg.values())
    print(f'{len(rows)} GEMM launches, {tot / 1e3:.2f} ms summed (each launch timed in isolation)')
    print(f'{"kind":7s} {"M":>6s} {"N":>5s} {"K*":>6s} {"taps":>4s} {"inst":>4s} {"tile":>7s} {"n":>3s} {"us each":>9s} '
          f'{"us total":>9s} {"TF":>6s}')
    for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        kind, M, N, K, taps, inst, var = key
        print(f'{kind:7s} {M:6d} {N:5d} {K:6d} {taps:4d} {inst:4d} {var:>7s} {a[0]:3d} {a[1] / a[0]:9.1f} {a[1]:9.1f} '
              f'{a[2] / a[1]:6.1f}')


if __name__ == '__main__':
    if len(sys.argv) >= 2 and sys.argv[1] == 'run':
        run(sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == '--model' else 'forward')
    elif len(sys.argv) >= 3 and sys.argv[1] == 'summarize':
        summarize(sys.argv[2])
    else:
        print(__doc__)
