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
for G, T, Hh, B in [(4, 841, 512, 32), (3, 841, 256, 32), (3, 128, 256, 32), (3, 128, 64, 32), (3, 128, 128, 32)]:
    xp = torch.randn(T, B, 2 * G * Hh, device=dev) * 0.1
    whh = [torch.randn(G * Hh, Hh, device=dev) * 0.03 for _ in range(2)]
    bhh = [torch.zeros(G * Hh, device=dev) for _ in range(2)]
    wt = [H.transpose2d(w) for w in whh]
    if G == 4:
        lens = torch.randint(T // 2, T + 1, (B,), device=dev); lens[0] = T
        f = lambda: H.lstm_fwd(xp, whh[0], whh[1], bhh[0], bhh[1], lens, Hh, True)
        raw, cst, gates = f(); dout = torch.randn_like(raw)
        b = lambda: H.lstm_bwd(dout, raw, cst, gates, wt[0], wt[1], lens, Hh)
    else:
        f = lambda: H.gru_fwd(xp, whh[0], whh[1], bhh[0], bhh[1], Hh, True)
        out, gates = f(); dout = torch.randn_like(out)
        b = lambda: H.gru_bwd(dout, out, gates, wt[0], wt[1], Hh)
    print(f'G{G} T{T} H{Hh}: fwd {timeit(f) * 1e3 / T:6.2f} us/step  bwd {timeit(b) * 1e3 / T:6.2f} us/step')
    H.check_rnn_status()
print('status ok')
