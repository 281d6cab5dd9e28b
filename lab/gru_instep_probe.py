"""Why does the postnet GRU-128 forward take 2.67 us/step inside the train step and 2.1 in isolation?  Times the same
launch (a) back to back with itself, (b) right behind a large GEMM, (c) with fresh output buffers every call vs reused."""
import sys, torch
sys.path.insert(0, '.')
from forwardtacotron_amd import hip as H
dev = 'cuda'
G, T, Hh, B = 3, 841, 128, 32
xp = torch.randn(T, B, 2 * G * Hh, device=dev) * 0.5
whh = [torch.randn(G * Hh, Hh, device=dev) * 0.1 for _ in range(2)]
bhh = [torch.zeros(G * Hh, device=dev) for _ in range(2)]
a = torch.randn(26912, 256, device=dev); w = torch.randn(512, 256, device=dev)
def gru(): return H.gru_fwd(xp, whh[0], whh[1], bhh[0], bhh[1], Hh, True)
def timed(pre):
    outs = []
    ts = []
    for _ in range(6):
        if pre: pre()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); outs.append(gru()); e.record(); ts.append((s, e))
    torch.cuda.synchronize()
    return [round(x.elapsed_time(y) * 1e3 / T, 2) for x, y in ts]
print('back to back      ', timed(None))
print('behind a GEMM     ', timed(lambda: H.linear_fwd(a, w, None)))
big = torch.empty(64 << 20, device=dev)
print('behind a 256MB fill', timed(lambda: big.fill_(1.0)))
xs = [torch.randn(T, B, 2 * G * Hh, device=dev) * 0.5 for _ in range(6)]
it = iter(xs)
def gru2():
    x = next(it)
    return H.gru_fwd(x, whh[0], whh[1], bhh[0], bhh[1], Hh, True)
ts = []
for _ in range(6):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); o = gru2(); e.record(); ts.append((s, e))
torch.cuda.synchronize()
print('fresh xp each call ', [round(x.elapsed_time(y) * 1e3 / T, 2) for x, y in ts])
H.check_rnn_status()
