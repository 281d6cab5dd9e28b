"""Phase timing of the persistent forward recurrence (needs a -DFT_RNN_TIMING build of ft_rnn.hip)."""
import sys, torch
sys.path.insert(0, '.')
from forwardtacotron_amd import hip as H
dev = 'cuda'
names = ['top(issue xg)', 'poll', 'barrier1', 'loads', 'mfma', 'partials+barrier2', 'cell', 'vmcnt0(stores)', 'barrier3+atomic']
for G, T, Hh, B in [(4, 841, 512, 32), (3, 841, 256, 32), (3, 128, 64, 32)]:
    xp = torch.randn(T, B, 2 * G * Hh, device=dev) * 0.1
    whh = [torch.randn(G * Hh, Hh, device=dev) * 0.03 for _ in range(2)]
    bhh = [torch.zeros(G * Hh, device=dev) for _ in range(2)]
    if G == 4:
        f = lambda: H.lstm_fwd(xp, whh[0], whh[1], bhh[0], bhh[1], None, Hh, True)
    else:
        f = lambda: H.gru_fwd(xp, whh[0], whh[1], bhh[0], bhh[1], Hh, True)
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); f(); e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e)
    ws, _ = H._rnn_workspace(G, B, Hh, dev)
    w = ws.view(torch.int32)[8:17].cpu().tolist()
    tot = sum(w)
    print(f'G{G} T{T} H{Hh}: {ms * 1e3 / T:.2f} us/step; ticks/16 total {tot} -> {ms * 1e3 / tot * 1000:.2f} ns per unit')
    for n, v in zip(names, w):
        print(f'   {n:22s} {v / tot * ms * 1e3 / T:6.3f} us/step  ({100 * v / tot:4.1f} %)')
H.check_rnn_status()
