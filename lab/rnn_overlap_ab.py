"""Recurrent layer forward: projection whole and in front (linear_multi_fwd + lstm_fwd / gru_fwd) vs in time chunks beside
the recurrence (ft_*_layer_fwd), at the benchmark's shapes and lengths.  ms per layer forward, HIP events on the stream."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from forwardtacotron_amd import data, hip as H  # noqa: E402


def timed(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    batch = data.synthetic_batch(B=32, Tmax=128, n_mels=80, seed=0)
    lens = batch['mel_len'].cuda()
    T = int(lens.max())
    B = 32
    g = torch.Generator().manual_seed(0)
    for G, I, Hh, packed in ((4, 512, 512, True), (3, 256, 256, False)):
        wih = [(torch.randn(G * Hh, I, generator=g) * 0.05).cuda() for _ in range(2)]
        whh = [(torch.randn(G * Hh, Hh, generator=g) * 0.04).cuda() for _ in range(2)]
        bih = [(torch.randn(G * Hh, generator=g) * 0.1).cuda() for _ in range(2)]
        bhh = [(torch.randn(G * Hh, generator=g) * 0.1).cuda() for _ in range(2)]
        x = torch.randn(B, T, I, generator=g).cuda()
        ln = lens if packed else None

        def seq():
            xp = H.linear_multi_fwd(x, wih, bih, y_tm_B=B)
            if G == 4:
                return H.lstm_fwd(xp, whh[0], whh[1], bhh[0], bhh[1], ln, Hh, True)
            return H.gru_fwd(xp, whh[0], whh[1], bhh[0], bhh[1], Hh, True)

        def rec_only(xp=H.linear_multi_fwd(x, wih, bih, y_tm_B=B)):
            if G == 4:
                return H.lstm_fwd(xp, whh[0], whh[1], bhh[0], bhh[1], ln, Hh, True)
            return H.gru_fwd(xp, whh[0], whh[1], bhh[0], bhh[1], Hh, True)

        print(f'G{G} T{T} H{Hh}: projection in front {timed(seq):.3f} ms (recurrence alone {timed(rec_only):.3f})', flush=True)
        for nch in (2, 4, 8, 12, 16):
            for lead in ((0, nch // 4, nch // 2, 3 * nch // 4) if packed else (0,)):
                os.environ['FT_RNN_REV_LEAD'] = str(lead)

                def ov():
                    if G == 4:
                        return H.lstm_layer_fwd(x, wih[0], wih[1], bih[0], bih[1], whh[0], whh[1], bhh[0], bhh[1], ln,
                                                Hh, True, nch)
                    return H.gru_layer_fwd(x, wih[0], wih[1], bih[0], bih[1], whh[0], whh[1], bhh[0], bhh[1], Hh, True,
                                           nch)
                print(f'   chunks {nch:2d} rev_lead {lead:2d}: {timed(ov):.3f} ms', flush=True)
        H.check_rnn_status()


if __name__ == '__main__':
    main()
