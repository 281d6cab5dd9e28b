"""Phase profile of the persistent recurrences (s_memtime stamps; diagnostic library built by lab/build_prof.sh).
    FT_LIB=lab/libfwdtaco_prof.so python lab/rnn_phase_prof.py [lstm_bwd|lstm_fwd|gru_bwd|gru_fwd]"""
import ctypes, os, sys, torch
sys.path.insert(0, '.')
from forwardtacotron_amd import hip as H, _lib
what = sys.argv[1] if len(sys.argv) > 1 else 'lstm_bwd'
G, T, Hh, B = (4, 841, 512, 32) if what.startswith('lstm') else (3, 841, 256, 32)
dev = 'cuda'
xp = torch.randn(T, B, 2 * G * Hh, device=dev) * 0.1
whh = [torch.randn(G * Hh, Hh, device=dev) * 0.03 for _ in range(2)]
bhh = [torch.zeros(G * Hh, device=dev) for _ in range(2)]
dout = torch.randn(T, B, 2 * Hh, device=dev) * 0.1
wt = [H.transpose2d(w) for w in whh]
if G == 4:
    raw, cst, gates = H.lstm_fwd(xp, whh[0], whh[1], bhh[0], bhh[1], None, Hh, True)
    fw = lambda: H.lstm_fwd(xp, whh[0], whh[1], bhh[0], bhh[1], None, Hh, True)
    bw = lambda: H.lstm_bwd(dout, raw, cst, gates, wt[0], wt[1], None, Hh)
else:
    out, gates = H.gru_fwd(xp, whh[0], whh[1], bhh[0], bhh[1], Hh, True)
    fw = lambda: H.gru_fwd(xp, whh[0], whh[1], bhh[0], bhh[1], Hh, True)
    bw = lambda: H.gru_bwd(dout, out, gates, wt[0], wt[1], Hh)
f = bw if what.endswith('bwd') else fw
for _ in range(3):
    f()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); f(); e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 1e3 / T
L = _lib.lib()
buf = (ctypes.c_ulonglong * 64)()
L.ft_rnn_prof_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.ft_rnn_prof_read(buf, 64) == 0
print(f'{what}: {us:.2f} us/step (instrumented)')
for slot in range(2):
    v = [buf[slot * 12 + i] / T for i in range(12)]
    tot = sum(v)
    print(f'  wave slot {slot}: cycles/step by phase ' + ' '.join(f'{x:7.0f}' for x in v) + f' | total {tot:.0f} cycles = {tot / us / 1e3:.2f} GHz-equivalent')
