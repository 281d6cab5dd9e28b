"""us/step of the persistent recurrences, forward and backward (events around one launch each)."""
import sys, torch
sys.path.insert(0, '.')
from forwardtacotron_amd import hip as H
dev = 'cuda'
def t(f):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); f(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3
for G, T, Hh, B in [(4, 841, 512, 32), (3, 841, 128, 32), (3, 128, 256, 32), (3, 128, 64, 32)]:
    xp = torch.randn(T, B, 2 * G * Hh, device=dev) * 0.1
    whh = [torch.randn(G * Hh, Hh, device=dev) * 0.03 for _ in range(2)]
    bhh = [torch.zeros(G * Hh, device=dev) for _ in range(2)]
    dout = torch.randn(T, B, 2 * Hh, device=dev) * 0.1
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
    print(f'G{G} T{T} H{Hh} B{B}: fwd {t(fw) / T:.2f} us/step, bwd {t(bw) / T:.2f} us/step')
H.check_rnn_status()
