"""GPU: mel inversion + Griffin-Lim (forwardtacotron_amd/vocoder.py, csrc/ft_dsp.hip; reference utils/dsp.py:80-94) against
the numpy oracle (oracle/gl_oracle.py) step by step.  PARITY UNPINNED against the reference itself: librosa is not
installed in this image and the reference ships no audio fixture (see the oracle's header) -- the oracle restates
librosa's published algorithm through FFTs, the product computes it through DFT-matrix GEMMs on the MFMA kernels."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = {'dsp': dict(num_mels=80, sample_rate=22050, hop_length=256, win_length=1024, n_fft=1024, fmin=0, fmax=8000,
                   peak_norm=False, trim_start_end_silence=True, trim_silence_top_db=60, trim_long_silences=False,
                   vad_window_length=30, vad_moving_average_width=8, vad_max_silence_length=12, vad_sample_rate=16000)}


def _signal(n, seed=0):
    t = np.arange(n) / 22050.0
    rng = np.random.default_rng(seed)
    return (0.5 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 1200 * t) * np.exp(-3 * t)
            + 0.05 * rng.standard_normal(n)).astype(np.float32)


@pytest.fixture(scope='module')
def gl():
    from forwardtacotron_amd.vocoder import GriffinLim
    return GriffinLim.from_config(CFG)          # the reference's DSP.from_config signature (utils/dsp.py:50-52)


def _split(X, Fp):
    """oracle complex [F, N] -> product layout [N, 2Fp]"""
    out = np.zeros((X.shape[1], 2 * Fp), dtype=np.float32)
    out[:, :X.shape[0]] = X.real.T
    out[:, Fp:Fp + X.shape[0]] = X.imag.T
    return out


def test_stft_and_istft_match_the_fft_oracle(gl):
    from oracle import gl_oracle as G
    y = _signal(256 * 120)
    X = G.stft(y, 1024, 256, 1024)
    got = gl.stft(torch.from_numpy(y).cuda()).cpu().numpy()
    assert got.shape == (X.shape[1], 2 * gl.Fp)
    scale = np.abs(X).max()
    assert np.abs(got - _split(X, gl.Fp)).max() < 2e-5 * scale            # fp32 DFT by GEMM, K = 1024
    ypad = gl.istft_padded(torch.from_numpy(_split(X, gl.Fp)).cuda()).cpu().numpy()
    want = G.istft(X, 256, 1024)
    assert ypad.shape == (1024 + 256 * (X.shape[1] - 1),)
    assert np.abs(ypad[512:-512] - want).max() < 2e-5 and float(np.abs(ypad[:512]).max()) == 0.0
    assert np.abs(want - y[:len(want)]).max() < 1e-6                      # (and the oracle inverts itself)


def test_mel_inversion_matches_the_oracle_iteration(gl):
    from oracle import gl_oracle as G
    S = np.abs(G.stft(_signal(256 * 90, seed=2), 1024, 256, 1024))
    B = G.mel_filterbank(22050, 1024, 80, 0, 8000)
    mel_log = np.log(np.clip(B @ S, 1e-5, None)).astype(np.float32)        # DSP.normalize (utils/dsp.py:96-98)
    want = G.mel_to_stft(np.exp(mel_log.astype(np.float64)), 22050, 1024, 0, 8000, nnls_iter=gl.nnls_iter)
    got = gl.mel_to_stft(torch.from_numpy(mel_log).cuda()).cpu().numpy()
    assert got.shape == (S.shape[1], gl.Fp) and float(got.min()) >= 0.0 and float(np.abs(got[:, 513:]).max()) == 0.0
    assert np.abs(got[:, :513].T - want).max() < 2e-4 * np.abs(want).max()
    Bm = B.astype(np.float64)
    assert np.linalg.norm(Bm @ got[:, :513].T - np.exp(mel_log)) / np.linalg.norm(np.exp(mel_log)) < 0.02


def test_griffinlim_matches_the_oracle_for_a_few_iterations_and_converges(gl):
    """same initial phases: after 3 iterations the waveforms agree to fp32 GEMM accuracy; after 32 (the reference's
    default n_iter) rounding differences have been amplified by the phase normalisation c / |c| at near-silent bins, so
    the 32-iteration check is the algorithm's own figure of merit (spectral convergence) against the oracle's."""
    from oracle import gl_oracle as G
    from forwardtacotron_amd.vocoder import spectral_convergence
    S = np.abs(G.stft(_signal(256 * 100, seed=3), 1024, 256, 1024))        # [513, 101]
    u = np.random.default_rng(7).random(S.shape)
    Sp = np.zeros((S.shape[1], gl.Fp), dtype=np.float32); Sp[:, :513] = S.T
    up = np.zeros_like(Sp); up[:, :513] = u.T
    Sd, ud = torch.from_numpy(Sp).cuda(), torch.from_numpy(up).cuda()
    for n_iter, tol in ((0, 1e-4), (1, 2e-4), (3, 1e-3)):
        want = G.griffinlim(S, n_iter, 256, 1024, u)
        got = gl.griffinlim_from_stft(Sd, n_iter, init_u=ud).cpu().numpy()
        assert got.shape == want.shape == (256 * 100,)
        assert np.abs(got - want).max() < tol * max(1.0, np.abs(want).max()), n_iter
    w32 = gl.griffinlim_from_stft(Sd, 32, init_u=ud)
    sc_got = spectral_convergence(gl, w32, Sd)
    sc_want = G.spectral_convergence(G.griffinlim(S, 32, 256, 1024, u), S, 1024, 256, 1024)
    sc0 = spectral_convergence(gl, gl.griffinlim_from_stft(Sd, 0, init_u=ud), Sd)
    assert sc_got < 0.5 * sc0 and abs(sc_got - sc_want) < 0.02, (sc0, sc_got, sc_want)


def test_dsp_griffinlim_entry_on_a_generated_mel(gl):
    """gen_forward.py:109-116 end to end on the drop-in: generate() -> mel_post -> griffinlim(n_iter=32) -> wav"""
    from forwardtacotron_amd.model import ForwardTacotron
    from helpers import TINY
    torch.manual_seed(0)
    m = ForwardTacotron(**dict(TINY, n_mels=80)).cuda()
    gen = m.generate(torch.randint(1, 100, (1, 12)).cuda(), alpha=1.0)
    mel = gen['mel_post'].clamp(-11.5, 2.0)                                 # an untrained model's "log-mel"
    wav = gl.griffinlim(mel, n_iter=32, seed=0)
    T = mel.shape[-1]
    assert wav.shape == (256 * (T - 1),) and bool(torch.isfinite(wav).all())
    wav2 = gl.griffinlim(mel.squeeze(0).cpu().numpy(), n_iter=32, seed=0)   # numpy input, same seed: reproducible
    assert torch.equal(wav, wav2)
