import sys, torch
sys.path.insert(0, '.')
from forwardtacotron_amd import hip as H
dev = 'cuda'
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
T, Hh = 841, 512
for B in (32, 17, 2):
    xp = torch.randn(T, B, 8 * Hh, device=dev) * 0.1
    whh = [torch.randn(4 * Hh, Hh, device=dev) * 0.03 for _ in range(2)]
    bhh = [torch.zeros(4 * Hh, device=dev) for _ in range(2)]
    ms = timeit(lambda: H.lstm_fwd(xp, whh[0], whh[1], bhh[0], bhh[1], None, Hh, False))
    print(f'LSTM fwd B{B} T{T} H{Hh} nolens nogates: {ms * 1e3 / T:6.2f} us/step')
    ms = timeit(lambda: H.lstm_fwd(xp, whh[0], whh[1], bhh[0], bhh[1], None, Hh, True))
    print(f'LSTM fwd B{B} T{T} H{Hh} nolens gates  : {ms * 1e3 / T:6.2f} us/step')
    raw, cst, gates = H.lstm_fwd(xp, whh[0], whh[1], bhh[0], bhh[1], None, Hh, True)
    dout = torch.randn_like(raw)
    wt = [H.transpose2d(w) for w in whh]
    ms = timeit(lambda: H.lstm_bwd(dout, raw, cst, gates, wt[0], wt[1], None, Hh))
    print(f'LSTM bwd B{B}: {ms * 1e3 / T:6.2f} us/step')
