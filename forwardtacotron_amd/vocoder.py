"""Mel inversion + Griffin-Lim on the MI355X (BASELINE configs[0]'s vocoder leg).

Reference: utils/dsp.py:80-94 `DSP.griffinlim(mel, n_iter=32)` -- exp, librosa `mel_to_stft(power=1)`, librosa
`griffinlim(n_iter, hop_length, win_length)` -- called from gen_forward.py:109-116 on `gen['mel_post']`.

    gl = GriffinLim.from_config(config)             # config['dsp'] as in configs/singlespeaker.yaml:8-26
    wav = gl.griffinlim(gen['mel_post'][0])         # log-mel [n_mels, T] (device tensor or numpy) -> wav [hop*(T-1)]

MI355X-first: both transforms are GEMMs on the package's MFMA kernels.  The forward DFT of all frames is ONE
`ft_linear_fwd` whose A operand is the zero-padded signal itself read with a row stride of `hop` samples (the frames
overlap in memory, nothing is gathered), against a [2F, n_fft] basis that has the Hann window folded in; the inverse is
one GEMM against the [n_fft, 2F] inverse basis (window folded in again) followed by an ordered overlap-add gather.  The
phase update and the mel pseudo-inverse steps are small element-wise kernels (csrc/ft_dsp.hip).  32 iterations on a
~800-frame utterance are 64 GEMMs of 1.7 GFLOP.

PARITY UNPINNED against the reference: librosa is not installed in this image and the reference ships no audio
fixture.  The oracle is oracle/gl_oracle.py (numpy, FFT-based -- an independent route to the same published algorithm);
tests/test_gpu_vocoder.py compares step by step.  Differences from librosa that are ours, not the reference's:
  * mel inversion: librosa runs scipy's L-BFGS-B NNLS from the clipped least-squares solution; here the same start is
    followed by `nnls_iter` projected-gradient steps (GEMMs), see the oracle's header;
  * the random initial phases come from numpy's default_rng(seed) on the host (librosa: default_rng(random_state)).
"""
from typing import Any, Dict, Optional, Union

import numpy as np
import torch

from . import _lib
from . import hip as H


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    lin = f / f_sp
    return np.where(f >= 1000.0, 1000.0 / f_sp + np.log(np.maximum(f, 1e-10) / 1000.0) / (np.log(6.4) / 27.0), lin)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    return np.where(m >= 1000.0 / f_sp, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 1000.0 / f_sp)), f_sp * m)


def slaney_mel_basis(sr: int, n_fft: int, n_mels: int, fmin: float, fmax: float) -> np.ndarray:
    """Slaney-scale triangular filters with area normalisation (what librosa.filters.mel builds by default)"""
    freqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    pts = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    up = (freqs[None, :] - pts[:-2, None]) / (pts[1:-1] - pts[:-2])[:, None]
    down = (pts[2:, None] - freqs[None, :]) / (pts[2:] - pts[1:-1])[:, None]
    w = np.clip(np.minimum(up, down), 0, None)
    return w * (2.0 / (pts[2:] - pts[:-2]))[:, None]


class GriffinLim:
    def __init__(self, num_mels: int, sample_rate: int, hop_length: int, win_length: int, n_fft: int, fmin: float,
                 fmax: float, device: Union[str, torch.device] = 'cuda', nnls_iter: int = 64, momentum: float = 0.99,
                 **_unused) -> None:
        self.n_mels, self.sr, self.hop, self.win_length, self.n_fft = num_mels, sample_rate, hop_length, win_length, n_fft
        self.fmin, self.fmax, self.nnls_iter, self.momentum = fmin, fmax, nnls_iter, momentum
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _lib.FtError('GriffinLim runs on an MI355X (HIP) device only; there is no CPU fallback')
        if n_fft % 4 or hop_length % 4 or not 0 < hop_length <= n_fft or win_length > n_fft:
            raise _lib.FtError('GriffinLim: n_fft and hop_length must be multiples of 4, hop <= n_fft, win <= n_fft')
        F = n_fft // 2 + 1
        self.F, self.Fp = F, (F + 3) // 4 * 4
        n = np.arange(win_length)
        win = 0.5 - 0.5 * np.cos(2 * np.pi * n / win_length)                 # periodic Hann, centred in n_fft
        lpad = (n_fft - win_length) // 2
        self.window = np.pad(win, (lpad, n_fft - win_length - lpad))
        k = np.arange(n_fft)
        ang = 2 * np.pi * np.outer(np.arange(F), k) / n_fft                 # [F, n_fft]
        fwd = np.zeros((2 * self.Fp, n_fft))
        fwd[:F] = np.cos(ang) * self.window                                 # Re X_m =  sum_k w_k x_k cos
        fwd[self.Fp:self.Fp + F] = -np.sin(ang) * self.window               # Im X_m = -sum_k w_k x_k sin
        c = np.full(F, 2.0)
        c[0] = 1.0
        if n_fft % 2 == 0:
            c[-1] = 1.0
        inv = np.zeros((n_fft, 2 * self.Fp))
        inv[:, :F] = (np.cos(ang) * c[:, None]).T * self.window[:, None] / n_fft
        inv[:, self.Fp:self.Fp + F] = (-np.sin(ang) * c[:, None]).T * self.window[:, None] / n_fft
        B = slaney_mel_basis(sample_rate, n_fft, num_mels, fmin, fmax)      # [n_mels, F]
        Bp = np.zeros((num_mels, self.Fp))
        Bp[:, :F] = B
        pinv = np.zeros((self.Fp, num_mels))
        pinv[:F] = np.linalg.pinv(B)
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)
        self.w_fwd, self.w_inv = f32(fwd), f32(inv)
        self.mel_basis, self.mel_basis_t, self.mel_pinv = f32(Bp), f32(Bp.T), f32(pinv)
        self.inv_lip = float(1.0 / np.linalg.norm(B, 2) ** 2)
        self._inv_wss: Dict[int, torch.Tensor] = {}

    @classmethod
    def from_config(cls, config: Dict[str, Any], **kw) -> 'GriffinLim':
        return cls(**config['dsp'], **kw)

    # ------------------------------------------------------------------------------------------------
    def _wss(self, N: int) -> torch.Tensor:
        t = self._inv_wss.get(N)
        if t is None:
            wss = np.zeros(self.n_fft + self.hop * (N - 1))
            w2 = self.window ** 2
            for n in range(N):
                wss[n * self.hop:n * self.hop + self.n_fft] += w2
            inv = np.where(wss > np.finfo(np.float32).tiny, 1.0 / np.maximum(wss, 1e-300), 1.0)
            t = torch.from_numpy(inv.astype(np.float32)).to(self.device)
            self._inv_wss[N] = t
        return t

    def mel_to_stft(self, mel_log: torch.Tensor) -> torch.Tensor:
        """log-mel [n_mels, N] (device) -> magnitudes [N, Fp] (frames-major, columns >= F are zero)"""
        H._chk(mel_log, 'mel')
        C, N = mel_log.shape
        if C != self.n_mels:
            raise _lib.FtError(f'mel_to_stft: expected {self.n_mels} mel channels, got {C}')
        st = H._stream()
        M = torch.empty(N, C, device=self.device)
        _lib.call('ft_exp_transpose', mel_log.data_ptr(), M.data_ptr(), C, N, st)
        X = H.linear_fwd(M, self.mel_pinv, relu=True)                     # clipped least squares [N, Fp]
        for _ in range(self.nnls_iter):
            R = H.linear_fwd(X, self.mel_basis)                           # [N, n_mels]
            _lib.call('ft_sub', R.data_ptr(), M.data_ptr(), R.data_ptr(), R.numel(), st)
            G = H.linear_fwd(R, self.mel_basis_t)                         # [N, Fp]
            _lib.call('ft_nnls_step', X.data_ptr(), G.data_ptr(), self.inv_lip, X.numel(), st)
        return X

    def istft_padded(self, proj: torch.Tensor) -> torch.Tensor:
        """split spectrum [N, 2Fp] -> zero-padded signal [n_fft + hop*(N-1)] (the signal starts at n_fft // 2)"""
        N = proj.shape[0]
        frames = H.linear_fwd(proj, self.w_inv)                            # [N, n_fft], window folded in
        ypad = torch.empty(self.n_fft + self.hop * (N - 1), device=self.device)
        _lib.call('ft_overlap_add', frames.data_ptr(), self._wss(N).data_ptr(), ypad.data_ptr(), N, self.n_fft, self.hop,
                  H._stream())
        return ypad

    def stft_of_padded(self, ypad: torch.Tensor, N: int) -> torch.Tensor:
        """zero-padded signal -> split spectrum [N, 2Fp]: one GEMM over the overlapping frames (row stride = hop)"""
        out = torch.empty(N, 2 * self.Fp, device=self.device)
        _lib.call('ft_linear_fwd', ypad.data_ptr(), self.hop, self.w_fwd.data_ptr(), None, out.data_ptr(), 2 * self.Fp, N,
                  self.n_fft, 2 * self.Fp, 0, 0, 0, 0, H._stream())
        return out

    def stft(self, y: torch.Tensor) -> torch.Tensor:
        """y [L] -> split spectrum [1 + L // hop, 2Fp]"""
        H._chk(y, 'y')
        pad = self.n_fft // 2
        ypad = torch.zeros(y.numel() + 2 * pad, device=self.device)
        ypad[pad:pad + y.numel()] = y
        return self.stft_of_padded(ypad, 1 + y.numel() // self.hop)

    def griffinlim_from_stft(self, S: torch.Tensor, n_iter: int = 32, init_u: Optional[torch.Tensor] = None,
                             seed: Optional[int] = None) -> torch.Tensor:
        """S [N, Fp] magnitudes -> wav [hop*(N-1)].  init_u [N, Fp] in [0,1): the initial phases / (2 pi)."""
        H._chk(S, 'S')
        N, Fp = S.shape
        if Fp != self.Fp:
            raise _lib.FtError(f'griffinlim: expected {self.Fp} (padded) frequency columns, got {Fp}')
        if init_u is None:
            u = np.zeros((N, Fp), dtype=np.float32)
            u[:, :self.F] = np.random.default_rng(seed).random((self.F, N)).T       # drawn [F, N] like the reference
            init_u = torch.from_numpy(u).to(self.device)
        H._chk(init_u, 'init_u')
        st = H._stream()
        proj = torch.empty(N, 2 * Fp, device=self.device)
        tprev = torch.empty(N, 2 * Fp, device=self.device)
        _lib.call('ft_gl_init', init_u.data_ptr(), S.data_ptr(), proj.data_ptr(), N, Fp, st)
        alpha = self.momentum / (1.0 + self.momentum)
        for it in range(n_iter):
            ypad = self.istft_padded(proj)
            rebuilt = self.stft_of_padded(ypad, N)
            _lib.call('ft_gl_phase', rebuilt.data_ptr(), tprev.data_ptr(), S.data_ptr(), proj.data_ptr(), N, Fp, alpha,
                      int(it > 0), st)
        ypad = self.istft_padded(proj)
        pad = self.n_fft // 2
        return ypad[pad:ypad.numel() - pad].clone()

    def griffinlim(self, mel: Union[torch.Tensor, np.ndarray], n_iter: int = 32, seed: Optional[int] = None,
                   init_u: Optional[torch.Tensor] = None) -> torch.Tensor:
        """DSP.griffinlim (utils/dsp.py:80-94): log-mel [n_mels, T] -> wav (device tensor, hop*(T-1) samples)"""
        if isinstance(mel, np.ndarray):
            mel = torch.from_numpy(np.ascontiguousarray(mel, dtype=np.float32))
        mel = mel.to(self.device, torch.float32).contiguous()
        if mel.dim() == 3 and mel.shape[0] == 1:
            mel = mel[0]
        return self.griffinlim_from_stft(self.mel_to_stft(mel), n_iter, init_u, seed)


def spectral_convergence(gl: GriffinLim, wav: torch.Tensor, S: torch.Tensor) -> float:
    """|| |STFT(wav)| - S ||_F / ||S||_F on the device (diagnostic)"""
    X = gl.stft(wav)
    N = min(X.shape[0], S.shape[0])
    mag = torch.sqrt(X[:N, :gl.Fp] ** 2 + X[:N, gl.Fp:] ** 2)
    return float(torch.linalg.norm(mag - S[:N]) / torch.linalg.norm(S[:N]).clamp_min(1e-30))


__all__ = ['GriffinLim', 'slaney_mel_basis', 'spectral_convergence']
