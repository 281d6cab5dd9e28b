"""What does an event record cost the stream it is recorded on?  A chain of N back-to-back GEMM-sized kernels, with and
without an (otherwise unused) event record after each; and with a second stream waiting on each event."""
import time, torch
dev = 'cuda'
a = torch.randn(2048, 2048, device=dev); b = torch.randn(2048, 2048, device=dev)
side = torch.cuda.Stream()
def run(mode, n=200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        torch.mm(a, b)
        if mode >= 1:
            ev = torch.cuda.Event(); ev.record()
            if mode == 2:
                side.wait_event(ev)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
for mode, name in ((0, 'kernels only'), (1, '+ event record each'), (2, '+ record and a second stream waiting on it')):
    run(mode, 50)
    print(f'{name:45s} {run(mode):8.1f} us per kernel')
