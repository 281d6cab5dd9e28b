"""CPU ORACLE of the mel-inversion / Griffin-Lim leg -- TEST INFRASTRUCTURE ONLY (never imported by the product).

Reference: utils/dsp.py:80-94 `DSP.griffinlim(mel, n_iter=32)` = denormalize (exp) -> librosa.feature.inverse.mel_to_stft
(power=1, Slaney mel basis, non-negative least squares) -> librosa.core.griffinlim (fast Griffin-Lim, momentum 0.99,
random initial phases, hann window, centred frames, zero padding), called from gen_forward.py:109-116.

PARITY UNPINNED: librosa (requirements.txt:5, unpinned version) is not installed in this image and the reference holds
no audio fixture, so nothing here can be checked against the reference's own output.  What is restated is librosa's
PUBLISHED algorithm (0.10 documentation / paper: Perraudin, Balazs, Sondergaard 2013, "A fast Griffin-Lim algorithm"):
  * `mel_filterbank`: Slaney mel scale (linear below 1 kHz, log above, 200/3 Hz per mel), triangular filters between
    n_mels + 2 equally spaced mel points, each scaled by 2 / (its bandwidth in Hz)  [librosa.filters.mel, htk=False,
    norm='slaney'];
  * `stft` / `istft`: periodic Hann window of win_length centred in n_fft, frames centred on t * hop with ZERO padding
    of n_fft // 2 on both sides, rfft; inverse = irfft, window again, overlap-add, division by the summed squared
    window where it exceeds the smallest normal float, n_fft // 2 samples cut from both ends;
  * `griffinlim`: angles_0 = exp(2 pi i u), u ~ U[0,1); per iteration  y = istft(S angles); R = stft(y);
    angles = R - momentum / (1 + momentum) * R_prev; angles /= |angles| + tiny; finally istft(S angles).
  * `mel_to_stft`: librosa solves min ||B X - M||, X >= 0 with L-BFGS-B from the clipped least-squares solution.  That
    solver is a scipy routine, not an algorithm of the reference; restated here -- and built on the GPU -- as what it
    computes: the SAME starting point followed by a fixed number of projected-gradient steps with step 1 / ||B||_2^2
    (monotone for this convex problem).  The product (forwardtacotron_amd/vocoder.py) runs exactly this iteration, so
    oracle and product are comparable step by step; neither is bit-comparable with L-BFGS-B.

Independent of the product: transforms here go through numpy's FFT, the product's through DFT-matrix GEMMs on MFMA.
"""
import numpy as np

TINY32 = float(np.finfo(np.float32).tiny)


def hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """[n_mels, 1 + n_fft // 2] float32"""
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, fftfreqs.size))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def hann_padded(win_length, n_fft):
    n = np.arange(win_length)
    w = 0.5 - 0.5 * np.cos(2 * np.pi * n / win_length)          # periodic ("fftbins") Hann
    lpad = (n_fft - win_length) // 2
    return np.pad(w, (lpad, n_fft - win_length - lpad))


def stft(y, n_fft, hop, win_length):
    """y [L] -> complex [1 + n_fft//2, 1 + L // hop]"""
    win = hann_padded(win_length, n_fft)
    yp = np.pad(np.asarray(y, dtype=np.float64), n_fft // 2)
    n_frames = 1 + (len(yp) - n_fft) // hop
    frames = np.stack([yp[t * hop:t * hop + n_fft] * win for t in range(n_frames)], axis=1)
    return np.fft.rfft(frames, axis=0)


def window_sumsquare(n_frames, n_fft, hop, win_length):
    win = hann_padded(win_length, n_fft) ** 2
    out = np.zeros(n_fft + hop * (n_frames - 1))
    for t in range(n_frames):
        out[t * hop:t * hop + n_fft] += win
    return out


def istft(X, hop, win_length):
    """complex [F, N] -> y [hop * (N - 1)]"""
    n_fft = 2 * (X.shape[0] - 1)
    n_frames = X.shape[1]
    win = hann_padded(win_length, n_fft)
    frames = np.fft.irfft(X, n=n_fft, axis=0) * win[:, None]
    y = np.zeros(n_fft + hop * (n_frames - 1))
    for t in range(n_frames):
        y[t * hop:t * hop + n_fft] += frames[:, t]
    wss = window_sumsquare(n_frames, n_fft, hop, win_length)
    nz = wss > TINY32
    y[nz] /= wss[nz]
    return y[n_fft // 2:len(y) - n_fft // 2]


def nnls_projected_gradient(B, M, n_iter):
    """min ||B X - M||_F, X >= 0.  B [n_mels, F], M [n_mels, N] -> X [F, N].  Start: clipped least squares (librosa's
    x0); then n_iter steps X <- max(0, X - B^T (B X - M) / L), L = ||B||_2^2."""
    B = np.asarray(B, dtype=np.float64)
    M = np.asarray(M, dtype=np.float64)
    X = np.maximum(np.linalg.pinv(B) @ M, 0)
    L = np.linalg.norm(B, 2) ** 2
    for _ in range(n_iter):
        X = np.maximum(X - (B.T @ (B @ X - M)) / L, 0)
    return X


def mel_to_stft(mel_lin, sr, n_fft, fmin, fmax, nnls_iter=64):
    """linear-amplitude mel [n_mels, N] -> magnitude spectrogram [F, N] (power = 1)"""
    B = mel_filterbank(sr, n_fft, mel_lin.shape[0], fmin, fmax)
    return nnls_projected_gradient(B, mel_lin, nnls_iter)


def griffinlim(S, n_iter, hop, win_length, init_u, momentum=0.99):
    """S [F, N] magnitudes; init_u [F, N] uniform numbers in [0,1) (the random initial phases, 2 pi u)"""
    angles = np.exp(2j * np.pi * np.asarray(init_u, dtype=np.float64))
    n_fft = 2 * (S.shape[0] - 1)
    tprev = None
    for _ in range(n_iter):
        inverse = istft(S * angles, hop, win_length)
        rebuilt = stft(inverse, n_fft, hop, win_length)
        angles = rebuilt.copy()
        if tprev is not None:
            angles -= (momentum / (1 + momentum)) * tprev
        angles /= np.abs(angles) + TINY32
        tprev = rebuilt
    return istft(S * angles, hop, win_length)


def dsp_griffinlim(mel_log, cfg, init_u, n_iter=32, nnls_iter=64):
    """DSP.griffinlim (utils/dsp.py:80-94): log-mel [n_mels, N] -> wav"""
    S = mel_to_stft(np.exp(np.asarray(mel_log, dtype=np.float64)), cfg['sample_rate'], cfg['n_fft'], cfg['fmin'],
                    cfg['fmax'], nnls_iter)
    return griffinlim(S, n_iter, cfg['hop_length'], cfg['win_length'], init_u)


def spectral_convergence(y, S, n_fft, hop, win_length):
    """|| |STFT(y)| - S ||_F / ||S||_F"""
    R = np.abs(stft(y, n_fft, hop, win_length))
    n = min(R.shape[1], S.shape[1])
    return float(np.linalg.norm(R[:, :n] - S[:, :n]) / max(np.linalg.norm(S[:, :n]), 1e-30))
