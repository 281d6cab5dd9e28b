"""weight-gradient (TN) GEMMs in isolation: FT_GEMM_TN_PIPE=0 (two-barrier kernel) vs 1 (software-pipelined) (one process per setting)."""
import os, subprocess, sys
sys.path.insert(0, '.')
def child():
    import torch
    from forwardtacotron_amd import hip as H
    def timeit(fn, n=10):
        fn(); fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n
    torch.manual_seed(0)
    for O, I, R in [(2048, 512, 26912), (2048, 1024, 26912), (256, 2048, 26912), (256, 256, 26912), (512, 256, 4096), (4096, 4096, 4096)]:
        dy = torch.randn(R, O, device='cuda'); x = torch.randn(R, I, device='cuda')
        dw = H.linear_bwd_weight(dy, x)
        ms = timeit(lambda: H.linear_bwd_weight(dy, x))
        print(f'TN {O:6d} {I:5d} {R:5d}: {ms*1e3:8.1f} us {2*O*I*R/ms/1e9:7.1f} TF  checksum {dw.double().abs().sum().item():.10e}', flush=True)
if __name__ == '__main__':
    if len(sys.argv) > 1: child()
    else:
        for v in ('0', '1', '0', '1'):
            print('=== FT_GEMM_TN_PIPE=' + v, flush=True)
            subprocess.run([sys.executable, __file__, 'child'], env=dict(os.environ, FT_GEMM_TN_PIPE=v), check=True)
