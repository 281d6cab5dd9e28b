"""A/B of the 128x128 NT split-GEMM kernels: FT_GEMM_PIPE=0 (two-barrier) vs 1 (software-pipelined), one process per
setting (the knob is read once).  usage: python lab/gemm_pipe_ab.py   (spawns itself twice)"""
import os, subprocess, sys
sys.path.insert(0, '.')

SHAPES_LIN = [(26912, 2048, 512), (26912, 2048, 1024), (26912, 512, 4096), (26912, 256, 512), (26912, 768, 512),
              (4096, 4096, 768), (26912, 1024, 80), (26912, 80, 1024), (8192, 8192, 1024), (26912, 2048, 520)]
SHAPES_CONV = [(32, 841, 80, 256, 8), (32, 841, 2048, 256, 3), (32, 841, 256, 80, 3), (32, 128, 256, 256, 16), (32, 841, 512, 256, 3)]


def child():
    import torch
    from forwardtacotron_amd import hip as H
    dev = 'cuda'

    def timeit(fn, n=10):
        fn(); fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n
    torch.manual_seed(0)
    for M, N, K in SHAPES_LIN:
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev)
        y = H.linear_fwd(x, w)
        ref = (x[:512].double() @ w.double().t())
        err = (y[:512].double() - ref).abs().max().item() / ref.abs().max().item()
        ms = timeit(lambda: H.linear_fwd(x, w))
        print(f'NT   {M:6d} {N:5d} {K:5d}: {ms*1e3:8.1f} us {2*M*N*K/ms/1e9:7.1f} TF  relerr {err:.2e}', flush=True)
    for B, T, Cin, Cout, k in SHAPES_CONV:
        x = torch.randn(B, T, Cin, device=dev); wp = torch.randn(k, Cout, Cin, device=dev)
        y = H.conv1d_fwd(x, wp, relu=False)
        # reference on a slice: torch conv1d ('same' padding as the reference's: left = (k-1)//2 ... checked by the parity tests)
        ms = timeit(lambda: H.conv1d_fwd(x, wp, relu=False))
        print(f'conv {B} {T} {Cin} {Cout} {k}: {ms*1e3:8.1f} us {2*B*T*Cin*Cout*k/ms/1e9:7.1f} TF  sum {y.double().sum().item():.6e} abs {y.double().abs().sum().item():.8e}', flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        child()
    else:
        for v in ('0', '1'):
            print(f'=== FT_GEMM_PIPE={v}', flush=True)
            env = dict(os.environ, FT_GEMM_PIPE=v)
            subprocess.run([sys.executable, __file__, 'child'], env=env, check=True)
