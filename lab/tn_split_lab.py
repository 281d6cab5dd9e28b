"""Split-K sweep of the weight-gradient GEMMs (run once per FT_TN_FORCE_S value): bank launch at the prenet and
postnet shapes, the LSTM W_hh gradient and a 3-tap conv, timed with HIP events (GEMM + its reduction)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from forwardtacotron_amd import hip as H

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

S = os.environ.get('FT_TN_FORCE_S', 'auto')
def bank(B, T, Cin, C, K):
    x = torch.randn(B, T, Cin, device='cuda'); dy = torch.randn(B, T + 1, K * C, device='cuda')
    dws = [torch.empty(C, Cin, k, device='cuda') for k in range(1, K + 1)]
    us = timeit(lambda: H.conv_bank_bwd_weight(dy, x, dws, C))
    fl = 2.0 * C * Cin * B * (T + 1) * K * (K + 1) / 2
    print(f'S={S:>4} bank B{B} T{T} Cin{Cin} C{C} K{K}: {us:8.1f} us {fl / us / 1e6:6.1f} TF')
def lin(rows, in_f, out_f):
    x = torch.randn(rows, in_f, device='cuda'); dy = torch.randn(rows, out_f, device='cuda')
    us = timeit(lambda: H.linear_bwd_weight(dy, x))
    print(f'S={S:>4} linear rows{rows} in{in_f} out{out_f}: {us:8.1f} us {2.0 * rows * in_f * out_f / us / 1e6:6.1f} TF')
bank(32, 128, 256, 256, 16)
bank(32, 841, 80, 256, 8)
lin(26912, 512, 2048)
lin(26912, 256, 768)
lin(4096, 4096, 256)
def conv(B, T, Cin, Cout, k):
    x = torch.randn(B, T, Cin, device='cuda'); dy = torch.randn(B, T + 1, Cout, device='cuda')
    dw = torch.empty(Cout, Cin, k, device='cuda')
    us = timeit(lambda: H.conv1d_bwd_weight_raw(dy.data_ptr(), Cout, x, dw, T + 1, T + 1))
    print(f'S={S:>4} conv B{B} T{T} Cin{Cin} Cout{Cout} k{k}: {us:8.1f} us {2.0 * Cout * Cin * B * (T + 1) * k / us / 1e6:6.1f} TF')
conv(32, 128, 256, 4096, 8)
conv(32, 128, 256, 2048, 16)
conv(32, 128, 256, 256, 16)
