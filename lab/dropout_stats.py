"""Statistics of the counter-based dropout mask (ft_dropout_keep): keep rate, adjacent-element and cross-seed correlation,
keep rate per residue class of the index (strided access patterns)."""
import sys
sys.path.insert(0, '.')
import torch
from forwardtacotron_amd import hip as H
n = 1 << 24
ones = torch.ones(n, device='cuda')
for p in (0.1, 0.5):
    ms = []
    for seed in (1, 2, 0x123456789abcdef, (1 << 61) + 12345):
        m = (H.dropout(ones, p, seed) != 0).float()
        ms.append(m)
        adj = float((m[1:] * m[:-1]).mean() - m.mean() ** 2)
        strides = [float(m[r::841].mean()) for r in (0, 1, 7)] + [float(m[r::4096].mean()) for r in (0, 5)]
        print(f'p={p} seed={seed:#x}: keep {float(m.mean()):.5f} (want {1 - p:.5f}), adjacent covariance {adj:+.2e}, '
              f'strided keep rates {["%.4f" % s for s in strides]}')
    cross = float((ms[0] * ms[1]).mean() - ms[0].mean() * ms[1].mean())
    print(f'p={p}: covariance between seeds 1 and 2: {cross:+.2e}  (std of an independent estimate ~ {(p * (1 - p)) / n ** 0.5:.1e})')
