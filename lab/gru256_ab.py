"""forward / backward GRU, H = 256, T = 841, B = 32 (the postnet's): workgroup-shape variants
(FT_RNN_GRU_WIDE for the forward: 8 waves x 1 k-block x 16 units against 4 x 2 x 8)."""
import os, subprocess, sys
sys.path.insert(0, '.')
def child():
    import torch
    from forwardtacotron_amd import hip as H
    T, B, Hh = 841, 32, 256
    g = torch.Generator().manual_seed(0)
    xp = (torch.randn(T, B, 6 * Hh, generator=g) * 0.2).cuda()
    whh = [(torch.randn(3 * Hh, Hh, generator=g) * 0.04).cuda() for _ in range(2)]
    bhh = [(torch.randn(3 * Hh, generator=g) * 0.05).cuda() for _ in range(2)]
    dout = (torch.randn(T, B, 2 * Hh, generator=g) * 0.1).cuda()
    f = lambda: H.gru_fwd(xp, whh[0], whh[1], bhh[0], bhh[1], Hh, True)
    out, gates = f(); torch.cuda.synchronize()
    wt = [H.transpose2d(w) for w in whh]
    b = lambda: H.gru_bwd(dout, out, gates, wt[0], wt[1], Hh)
    dxp, dhp = b(); torch.cuda.synchronize()
    def best(fn):
        r = 1e9
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); fn(); e.record(); torch.cuda.synchronize()
            r = min(r, s.elapsed_time(e) * 1e3 / T)
        return r
    tf, tb = best(f), best(b)
    H.check_rnn_status()
    print(f'forward {tf:.3f} backward {tb:.3f} us/step  checksums {out.double().abs().sum().item():.10e} '
          f'{dxp.double().abs().sum().item():.10e} {dhp.double().abs().sum().item():.10e}  modes {H.rnn_mode_counts()}', flush=True)
if __name__ == '__main__':
    if len(sys.argv) > 1: child()
    else:
        for env in ({'FT_RNN_GRU_WIDE': '0'}, {'FT_RNN_GRU_WIDE': '1'}, {'FT_RNN_GRU_WIDE': '0'}, {'FT_RNN_GRU_WIDE': '1'}):
            print('===', env, flush=True)
            subprocess.run([sys.executable, __file__, 'child'], env=dict(os.environ, **env), check=True)
