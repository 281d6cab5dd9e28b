"""us/step of the persistent recurrences, forward and backward (events around one launch each), XCD-local hand-off
against the agent-scope one (FT_RNN_LOCAL toggled in-process), bit-equality of the two, and how many groups really ran
XCD-local.    python lab/rnn_step_us.py"""
import os, sys, torch
sys.path.insert(0, '.')
from forwardtacotron_amd import hip as H
dev = 'cuda'
def t(f, n=3):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); r = f(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3)
    return best, r
shapes = [(4, 841, 512, 32), (3, 841, 128, 32), (3, 128, 256, 32), (3, 128, 64, 32), (3, 128, 128, 32)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for G, T, Hh, B in shapes:
    xp = torch.randn(T, B, 2 * G * Hh, device=dev) * 0.1
    whh = [torch.randn(G * Hh, Hh, device=dev) * 0.03 for _ in range(2)]
    bhh = [torch.zeros(G * Hh, device=dev) for _ in range(2)]
    dout = torch.randn(T, B, 2 * Hh, device=dev) * 0.1
    res = {}
    for mode in ('1', '0'):
        os.environ['FT_RNN_LOCAL'] = mode
        m0 = H.rnn_mode_counts()
        if G == 4:
            raw, cst, gates = H.lstm_fwd(xp, whh[0], whh[1], bhh[0], bhh[1], None, Hh, True)
            fw = lambda: H.lstm_fwd(xp, whh[0], whh[1], bhh[0], bhh[1], None, Hh, True)
            wt = [H.transpose2d(w) for w in whh]
            bw = lambda: H.lstm_bwd(dout, raw, cst, gates, wt[0], wt[1], None, Hh)
        else:
            out, gates = H.gru_fwd(xp, whh[0], whh[1], bhh[0], bhh[1], Hh, True)
            fw = lambda: H.gru_fwd(xp, whh[0], whh[1], bhh[0], bhh[1], Hh, True)
            wt = [H.transpose2d(w) for w in whh]
            bw = lambda: H.gru_bwd(dout, out, gates, wt[0], wt[1], Hh)
        tf, rf = t(fw)
        tb, rb = t(bw)
        m1 = H.rnn_mode_counts()
        res[mode] = (tf / T, tb / T, rf, rb, (m1[0] - m0[0], m1[1] - m0[1]))
    a, b = res['1'], res['0']
    same = all(torch.equal(x, y) for x, y in zip([v for v in a[2] if v is not None], [v for v in b[2] if v is not None]))
    sameb = all(torch.equal(x, y) for x, y in zip(a[3] if isinstance(a[3], tuple) else (a[3],), b[3] if isinstance(b[3], tuple) else (b[3],)))
    print(f'G{G} T{T} H{Hh} B{B}: local fwd {a[0]:.2f} bwd {a[1]:.2f} us/step (groups local/agent {a[4]}) | '
          f'agent-scope fwd {b[0]:.2f} bwd {b[1]:.2f} {b[4]} | bit-equal fwd {same} bwd {sameb}', flush=True)
H.check_rnn_status()
print('persistent/refused', H.rnn_counters())
