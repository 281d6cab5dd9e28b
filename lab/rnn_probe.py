import sys, torch, time
sys.path.insert(0, '.')
from forwardtacotron_amd import hip as H
torch.manual_seed(0)
dev = 'cuda'
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for (B, T, Hh) in [(32, 841, 512), (32, 128, 512), (32, 841, 256)]:
    xp = torch.randn(T, B, 8 * Hh, device=dev) * 0.1
    whh = [torch.randn(4 * Hh, Hh, device=dev) * 0.03 for _ in range(2)]
    bhh = [torch.zeros(4 * Hh, device=dev) for _ in range(2)]
    lens = torch.randint(T // 2, T + 1, (B,), device=dev); lens[0] = T
    for name, kw in [('full', dict(lens=lens, save=True)), ('nogates', dict(lens=lens, save=False)),
                     ('nolens', dict(lens=None, save=True)), ('nolens_nogates', dict(lens=None, save=False))]:
        ms = timeit(lambda: H.lstm_fwd(xp, whh[0], whh[1], bhh[0], bhh[1], kw['lens'], Hh, kw['save']))
        print(f'LSTM fwd B{B} T{T} H{Hh} {name:16s}: {ms:8.3f} ms  = {ms * 1e3 / T:6.2f} us/step')
    raw, cst, gates = H.lstm_fwd(xp, whh[0], whh[1], bhh[0], bhh[1], lens, Hh, True)
    dout = torch.randn_like(raw)
    wt = [H.transpose2d(w) for w in whh]
    ms = timeit(lambda: H.lstm_bwd(dout, raw, cst, gates, wt[0], wt[1], lens, Hh))
    print(f'LSTM bwd B{B} T{T} H{Hh}: {ms:8.3f} ms = {ms * 1e3 / T:6.2f} us/step')
# empty-kernel launch rate reference: tiny scale kernel
x = torch.zeros(64, device=dev)
ms = timeit(lambda: [H.scale(x, 1.0) for _ in range(200)])
print(f'python-launched tiny kernel: {ms * 1e3 / 200:.2f} us each')
