import sys, torch
sys.path.insert(0, '.')
from forwardtacotron_amd import hip as H
dev = 'cuda'
def timeit(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
print('--- linear fwd (NT)  M,N,K')
for M, N, K in [(4096, 4096, 4096), (8192, 8192, 1024), (26912, 2048, 512), (26912, 512, 256), (4096, 512, 256), (4096, 768, 256), (26912, 80, 1024)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev)
    ms = timeit(lambda: H.linear_fwd(x, w))
    print(f'NT {M:6d} {N:5d} {K:5d}: {ms:8.3f} ms  {2*M*N*K/ms/1e9:7.1f} TF')
print('--- linear bwd data (NN)')
for M, N, K in [(4096, 4096, 4096), (26912, 512, 2048), (4096, 256, 512)]:
    dy = torch.randn(M, K, device=dev); w = torch.randn(K, N, device=dev)
    ms = timeit(lambda: H.linear_bwd_data(dy, w))
    print(f'NN {M:6d} {N:5d} {K:5d}: {ms:8.3f} ms  {2*M*N*K/ms/1e9:7.1f} TF')
print('--- linear bwd weight (TN)  out,in,rows')
for O, I, R in [(4096, 4096, 4096), (2048, 512, 26912), (256, 256, 26912), (512, 256, 4096), (256, 2048, 26912)]:
    dy = torch.randn(R, O, device=dev); x = torch.randn(R, I, device=dev)
    ms = timeit(lambda: H.linear_bwd_weight(dy, x))
    print(f'TN {O:6d} {I:5d} {R:5d}: {ms:8.3f} ms  {2*O*I*R/ms/1e9:7.1f} TF')
print('--- conv fwd  B,T,Cin,Cout,k')
for B, T, Cin, Cout, k in [(32, 841, 2048, 256, 3), (32, 841, 256, 80, 3), (32, 128, 4096, 256, 3), (32, 128, 256, 256, 5), (32, 841, 80, 256, 8)]:
    x = torch.randn(B, T, Cin, device=dev); wp = torch.randn(k, Cout, Cin, device=dev)
    ms = timeit(lambda: H.conv1d_fwd(x, wp, relu=True))
    print(f'conv {B} {T} {Cin} {Cout} {k}: {ms:8.3f} ms  {2*B*T*Cin*Cout*k/ms/1e9:7.1f} TF')
