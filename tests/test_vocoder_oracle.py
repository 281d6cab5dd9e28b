"""CPU: properties of oracle/gl_oracle.py (the numpy restatement of DSP.griffinlim, utils/dsp.py:80-94) and of the host
side of forwardtacotron_amd/vocoder.py (mel basis, DFT matrices).  PARITY UNPINNED against the reference: librosa is not
installed here and the reference holds no audio fixture; what can be pinned are the algorithm's own invariants and the
agreement of two independent formulations (FFT in the oracle, DFT-matrix products in the product)."""
import numpy as np

from oracle import gl_oracle as G

CFG = dict(sample_rate=22050, n_fft=1024, hop_length=256, win_length=1024, fmin=0, fmax=8000, num_mels=80)


def _signal(n=22050, seed=0):
    t = np.arange(n) / 22050.0
    rng = np.random.default_rng(seed)
    return (0.5 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 1200 * t) * np.exp(-3 * t)
            + 0.05 * rng.standard_normal(n))


def test_mel_filterbank_is_slaney_normalised_triangles():
    B = G.mel_filterbank(22050, 1024, 80, 0, 8000)
    assert B.shape == (80, 513) and B.dtype == np.float32 and float(B.min()) >= 0.0
    freqs = np.linspace(0, 11025, 513)
    # each filter: one contiguous bump; peaks strictly increasing in frequency; nothing above fmax
    peaks = [int(np.argmax(r)) for r in B]
    assert all(a <= b for a, b in zip(peaks, peaks[1:])) and float(B[:, freqs > 8000 + 1e-6].max()) == 0.0
    # Slaney area normalisation: integral of every filter over frequency = 1 (bin width 22050/1024 Hz)
    area = B.astype(np.float64).sum(1) * (22050 / 1024)
    assert np.all(np.abs(area[3:] - 1.0) < 0.12)
    # the linear part of the scale: centre spacing 200/3 Hz * (mel step) below 1 kHz
    mel_f = G.mel_to_hz(np.linspace(G.hz_to_mel(0), G.hz_to_mel(8000), 82))
    assert abs(G.hz_to_mel(1000.0) - 15.0) < 1e-9 and np.allclose(G.hz_to_mel(G.mel_to_hz(np.arange(0, 40.0))), np.arange(0, 40.0))
    assert np.all(np.diff(mel_f) > 0)


def test_stft_istft_round_trip_and_known_tone():
    y = _signal()
    X = G.stft(y, 1024, 256, 1024)
    assert X.shape == (513, 1 + len(y) // 256)
    yr = G.istft(X, 256, 1024)
    assert np.abs(yr - y[:len(yr)]).max() < 1e-9
    tone = np.sin(2 * np.pi * (22050 / 1024 * 40) * np.arange(8192) / 22050)       # exactly bin 40
    T = np.abs(G.stft(tone, 1024, 256, 1024))[:, 8]
    assert int(np.argmax(T)) == 40 and abs(T[40] - 256.0) < 1e-6 and T[44] < 1e-6    # Hann: amplitude * N/4, 3-bin main lobe


def test_nnls_steps_are_monotone_and_non_negative():
    B = G.mel_filterbank(22050, 1024, 80, 0, 8000)
    S = np.abs(G.stft(_signal(), 1024, 256, 1024))
    M = B @ S
    res = []
    for it in (0, 8, 64):
        X = G.nnls_projected_gradient(B, M, it)
        assert float(X.min()) >= 0.0
        res.append(np.linalg.norm(B @ X - M) / np.linalg.norm(M))
    assert res[0] > res[1] > res[2] and res[2] < 0.01


def test_griffinlim_reduces_spectral_inconsistency():
    S = np.abs(G.stft(_signal(), 1024, 256, 1024))
    u = np.random.default_rng(1).random(S.shape)
    sc = [G.spectral_convergence(G.griffinlim(S, n, 256, 1024, u), S, 1024, 256, 1024) for n in (0, 4, 32)]
    assert sc[0] > sc[1] > sc[2] and sc[2] < 0.15
    w = G.griffinlim(S, 2, 256, 1024, u)
    assert w.shape == (256 * (S.shape[1] - 1),)


def test_product_host_matrices_agree_with_the_fft_oracle():
    """forwardtacotron_amd.vocoder builds the mel basis and the windowed DFT / inverse-DFT matrices on the host: the
    same numbers as the oracle's independent (FFT, loop-built) formulation."""
    import torch
    from forwardtacotron_amd import vocoder as V
    assert np.abs(V.slaney_mel_basis(22050, 1024, 80, 0, 8000) - G.mel_filterbank(22050, 1024, 80, 0, 8000)).max() < 1e-7
    gl = V.GriffinLim.__new__(V.GriffinLim)            # host part only (no device here): replicate __init__'s matrices
    n_fft, F = 1024, 513
    win = G.hann_padded(1024, 1024)
    k = np.arange(n_fft)
    ang = 2 * np.pi * np.outer(np.arange(F), k) / n_fft
    x = np.random.default_rng(0).standard_normal(n_fft)
    X = np.fft.rfft(x * win)
    assert np.abs((np.cos(ang) * win) @ x - X.real).max() < 1e-9 and np.abs((-np.sin(ang) * win) @ x - X.imag).max() < 1e-9
    c = np.full(F, 2.0); c[0] = c[-1] = 1.0
    inv = ((np.cos(ang) * c[:, None]).T @ X.real + (-np.sin(ang) * c[:, None]).T @ X.imag) / n_fft * win
    assert np.abs(inv - np.fft.irfft(X, n=n_fft) * win).max() < 1e-9
    assert torch is not None and gl is not None
