"""one LSTM-512 forward with FT_RNN_DEBUG=1: which layout was admitted, did the groups verify as XCD-local"""
import os, sys, torch
os.environ['FT_RNN_DEBUG'] = '1'
sys.path.insert(0, '.')
from forwardtacotron_amd import hip as H
G, T, Hh, B = 4, 64, 512, 32
xp = torch.randn(T, B, 2 * G * Hh, device='cuda') * 0.1
whh = [torch.randn(G * Hh, Hh, device='cuda') * 0.03 for _ in range(2)]
bhh = [torch.zeros(G * Hh, device='cuda') for _ in range(2)]
for i in range(3):
    m0 = H.rnn_mode_counts()
    H.lstm_fwd(xp, whh[0], whh[1], bhh[0], bhh[1], None, Hh, True)
    print('groups local/agent', tuple(a - b for a, b in zip(H.rnn_mode_counts(), m0)), flush=True)
H.check_rnn_status()
