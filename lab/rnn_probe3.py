import sys, torch
sys.path.insert(0, '.')
from forwardtacotron_amd import hip as H
dev = 'cuda'
T, Hh, B = 200, 512, 32
xp = torch.randn(T, B, 8 * Hh, device=dev) * 0.1
whh = [torch.randn(4 * Hh, Hh, device=dev) * 0.03 for _ in range(2)]
bhh = [torch.zeros(4 * Hh, device=dev) for _ in range(2)]
raw, cst, gates = H.lstm_fwd(xp, whh[0], whh[1], bhh[0], bhh[1], None, Hh, True)
dout = torch.randn_like(raw)
wt = [H.transpose2d(w) for w in whh]
dg = H.lstm_bwd(dout, raw, cst, gates, wt[0], wt[1], None, Hh)
torch.cuda.synchronize()
