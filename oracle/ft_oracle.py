"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

Plain restatement (explicit tensor arithmetic on CPU, no nn.Module, no fused
torch RNN/conv/batchnorm operators) of the ForwardTacotron hot path of
ziyaad30/ForwardTacotron.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this file, and only as the
checker.  The product (``forwardtacotron_amd``) must never route through it.

Pinning: every function below is checked against outputs captured from the
imported reference (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``)
by ``tests/test_oracle_golden.py``; where /root/reference is present the same
test module also compares against the live import at full size.

Each function cites the reference file:line (relative to /root/reference) it
follows.  Parameters are addressed by the reference's ``state_dict`` key names.

All functions take/return torch CPU tensors; dtype follows the inputs so the
oracle can be run in float32 (parity) or float64 (error attribution).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

Tensor = torch.Tensor
PAD_VALUE = -11.5129  # models/forward_tacotron.py:69  (ln 1e-5, utils/dsp.py:96-98)
BN_EPS = 1e-5         # torch.nn.BatchNorm1d default, models/common_layers.py:51
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------- #
# primitives
# --------------------------------------------------------------------------- #
def embedding(idx: Tensor, weight: Tensor) -> Tensor:
    """nn.Embedding row gather. models/forward_tacotron.py:73,133 ; :18,31."""
    return weight[idx]


def conv1d(x: Tensor, w: Tensor, bias: Optional[Tensor] = None) -> Tensor:
    """nn.Conv1d(stride=1, padding=k//2).  models/common_layers.py:50.

    x [B,Cin,T], w [Cout,Cin,k] -> [B,Cout,T + 2*(k//2) - k + 1]
    out[b,co,t] = sum_j sum_ci w[co,ci,j] * xpad[b,ci,t+j], xpad zero-padded by
    k//2 on both sides (so even k yields T+1 outputs).
    """
    B, Cin, T = x.shape
    Cout, _, k = w.shape
    p = k // 2
    xpad = torch.zeros(B, Cin, T + 2 * p, dtype=x.dtype)
    xpad[:, :, p:p + T] = x
    Tout = T + 2 * p - k + 1
    out = torch.zeros(B, Cout, Tout, dtype=x.dtype)
    for j in range(k):
        out = out + torch.einsum('oc,bct->bot', w[:, :, j], xpad[:, :, j:j + Tout])
    if bias is not None:
        out = out + bias.view(1, -1, 1)
    return out


def batchnorm1d(x: Tensor, gamma: Tensor, beta: Tensor, running_mean: Tensor,
                running_var: Tensor, training: bool
                ) -> Tuple[Tensor, Tensor, Tensor]:
    """nn.BatchNorm1d over [B,C,T] (models/common_layers.py:51,57).

    Train: batch mean / biased variance over all B*T positions (padding
    included); running stats updated with momentum 0.1 and UNBIASED variance.
    Returns (y, new_running_mean, new_running_var).
    """
    if training:
        n = x.shape[0] * x.shape[2]
        mean = x.mean(dim=(0, 2))
        var = ((x - mean.view(1, -1, 1)) ** 2).mean(dim=(0, 2))
        unbiased = var * (n / max(n - 1, 1))
        new_rm = (1 - BN_MOMENTUM) * running_mean + BN_MOMENTUM * mean.detach()
        new_rv = (1 - BN_MOMENTUM) * running_var + BN_MOMENTUM * unbiased.detach()
    else:
        mean, var = running_mean, running_var
        new_rm, new_rv = running_mean, running_var
    y = (x - mean.view(1, -1, 1)) / torch.sqrt(var.view(1, -1, 1) + BN_EPS)
    y = y * gamma.view(1, -1, 1) + beta.view(1, -1, 1)
    return y, new_rm, new_rv


def batchnorm_conv(x: Tensor, P: Dict[str, Tensor], prefix: str, relu: bool,
                   training: bool, new_buffers: Optional[Dict[str, Tensor]] = None
                   ) -> Tensor:
    """BatchNormConv.forward: conv -> ReLU (if relu) -> BN.  common_layers.py:54-57."""
    y = conv1d(x, P[prefix + 'conv.weight'])
    if relu:
        y = torch.clamp_min(y, 0.)
    y, rm, rv = batchnorm1d(y, P[prefix + 'bnorm.weight'], P[prefix + 'bnorm.bias'],
                            P[prefix + 'bnorm.running_mean'],
                            P[prefix + 'bnorm.running_var'], training)
    if training and new_buffers is not None:
        new_buffers[prefix + 'bnorm.running_mean'] = rm
        new_buffers[prefix + 'bnorm.running_var'] = rv
        new_buffers[prefix + 'bnorm.num_batches_tracked'] = \
            P[prefix + 'bnorm.num_batches_tracked'] + 1
    return y


def maxpool_k2s1p1(x: Tensor) -> Tensor:
    """MaxPool1d(kernel 2, stride 1, padding 1)[:, :, :T]  (common_layers.py:78,105):
    out[t] = max(x[t-1], x[t]) with x[-1] = -inf.  torch's max_pool keeps the FIRST maximal element
    of a window (strict '>' update), which decides where the gradient goes on ties (post-ReLU zeros)."""
    prev = torch.full_like(x, -float('inf'))
    prev[:, :, 1:] = x[:, :, :-1]
    return torch.where(x > prev, x, prev)


def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    y = x @ w.t()
    return y + b if b is not None else y


def highway(x: Tensor, P: Dict[str, Tensor], prefix: str) -> Tensor:
    """HighwayNetwork.forward.  common_layers.py:35-40."""
    x1 = linear(x, P[prefix + 'W1.weight'], P[prefix + 'W1.bias'])
    x2 = linear(x, P[prefix + 'W2.weight'], P[prefix + 'W2.bias'])
    g = torch.sigmoid(x2)
    return g * torch.clamp_min(x1, 0.) + (1. - g) * x


def gru_direction(x: Tensor, w_ih: Tensor, w_hh: Tensor, b_ih: Tensor, b_hh: Tensor,
                  reverse: bool) -> Tensor:
    """One direction of nn.GRU(batch_first) over the full padded length, h0 = 0.
    Gate order r,z,n; n = tanh(W_in x + b_in + r*(W_hn h + b_hn));
    h' = (1-z)*n + z*h.   (common_layers.py:89,123 ; forward_tacotron.py:24,37)
    """
    B, T, _ = x.shape
    H = w_hh.shape[1]
    xp = x @ w_ih.t() + b_ih                     # [B,T,3H]
    h = torch.zeros(B, H, dtype=x.dtype)
    outs: List[Optional[Tensor]] = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        hp = h @ w_hh.t() + b_hh
        r = torch.sigmoid(xp[:, t, 0:H] + hp[:, 0:H])
        z = torch.sigmoid(xp[:, t, H:2 * H] + hp[:, H:2 * H])
        n = torch.tanh(xp[:, t, 2 * H:] + r * hp[:, 2 * H:])
        h = (1. - z) * n + z * h
        outs[t] = h
    return torch.stack(outs, dim=1)


def bigru(x: Tensor, P: Dict[str, Tensor], prefix: str) -> Tensor:
    f = gru_direction(x, P[prefix + 'weight_ih_l0'], P[prefix + 'weight_hh_l0'],
                      P[prefix + 'bias_ih_l0'], P[prefix + 'bias_hh_l0'], False)
    r = gru_direction(x, P[prefix + 'weight_ih_l0_reverse'], P[prefix + 'weight_hh_l0_reverse'],
                      P[prefix + 'bias_ih_l0_reverse'], P[prefix + 'bias_hh_l0_reverse'], True)
    return torch.cat([f, r], dim=-1)


def lstm_direction(x: Tensor, lens: Optional[Tensor], w_ih: Tensor, w_hh: Tensor,
                   b_ih: Tensor, b_hh: Tensor, reverse: bool, pad_value: float) -> Tensor:
    """One direction of nn.LSTM(batch_first), h0=c0=0, gate order i,f,g,o.

    With ``lens`` (pack_padded_sequence / pad_packed_sequence semantics,
    forward_tacotron.py:147-152): item b is processed over exactly lens[b]
    frames (the reverse direction starts at frame lens[b]-1) and output frames
    t >= lens[b] are filled with ``pad_value``.  ``lens=None`` = run over the
    whole padded length (generate path, forward_tacotron.py:224).
    """
    B, T, _ = x.shape
    H = w_hh.shape[1]
    xp = x @ w_ih.t() + b_ih + b_hh
    h = torch.zeros(B, H, dtype=x.dtype)
    c = torch.zeros(B, H, dtype=x.dtype)
    outs: List[Optional[Tensor]] = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        g = xp[:, t] + h @ w_hh.t()
        i_ = torch.sigmoid(g[:, 0:H])
        f_ = torch.sigmoid(g[:, H:2 * H])
        g_ = torch.tanh(g[:, 2 * H:3 * H])
        o_ = torch.sigmoid(g[:, 3 * H:])
        c_new = f_ * c + i_ * g_
        h_new = o_ * torch.tanh(c_new)
        if lens is None:
            h, c = h_new, c_new
            outs[t] = h_new
        else:
            act = (t < lens).to(x.dtype).view(B, 1)
            h = act * h_new + (1. - act) * h
            c = act * c_new + (1. - act) * c
            outs[t] = act * h_new + (1. - act) * pad_value
    return torch.stack(outs, dim=1)


def bilstm(x: Tensor, lens: Optional[Tensor], P: Dict[str, Tensor], prefix: str,
           pad_value: float = PAD_VALUE) -> Tensor:
    """With ``lens``: pad_packed_sequence returns max(lens) frames (forward_tacotron.py:151), so an input expanded to
    more frames than that is cut there (the packed LSTM never reads the rest); an item longer than the input makes the
    reference's packed LSTM raise -- so does this."""
    if lens is not None:
        Lmax = int(lens.max())
        if Lmax > x.shape[1]:
            raise RuntimeError(f'packed length {Lmax} exceeds the {x.shape[1]} frames of the input')
        x = x[:, :Lmax]
    f = lstm_direction(x, lens, P[prefix + 'weight_ih_l0'], P[prefix + 'weight_hh_l0'],
                       P[prefix + 'bias_ih_l0'], P[prefix + 'bias_hh_l0'], False, pad_value)
    r = lstm_direction(x, lens, P[prefix + 'weight_ih_l0_reverse'],
                       P[prefix + 'weight_hh_l0_reverse'], P[prefix + 'bias_ih_l0_reverse'],
                       P[prefix + 'bias_hh_l0_reverse'], True, pad_value)
    return torch.cat([f, r], dim=-1)


# --------------------------------------------------------------------------- #
# LengthRegulator -- integer index arithmetic in numpy, bit-exact row copies
# --------------------------------------------------------------------------- #
def lr_repeats(dur: np.ndarray) -> np.ndarray:
    """common_layers.py:18,21: dur[dur<0]=0 ; r = (dur + 0.5).long()  (trunc toward 0)."""
    d = np.where(dur < 0, np.float32(0), dur).astype(np.float32)
    return np.trunc((d + np.float32(0.5)).astype(np.float32)).astype(np.int64)


def lr_index_map(dur: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Returns (src [B,Tm] int64 token index per output frame or -1 for padding,
    total [B] frames per item); Tm = max_b total[b]."""
    r = lr_repeats(dur)
    total = r.sum(axis=1)
    Tm = int(total.max()) if total.size else 0
    src = -np.ones((dur.shape[0], Tm), dtype=np.int64)
    for b in range(dur.shape[0]):
        pos = 0
        for j in range(dur.shape[1]):
            n = int(r[b, j])
            src[b, pos:pos + n] = j
            pos += n
    return src, total


def length_regulate(x: Tensor, dur: Tensor) -> Tensor:
    """LengthRegulator.forward (common_layers.py:17-24), differentiable in x.
    Mutates ``dur`` in place (dur[dur<0]=0) exactly like the reference."""
    dur[dur < 0] = 0.
    src, _ = lr_index_map(dur.detach().cpu().numpy().astype(np.float32))
    B, Tm = src.shape
    src_t = torch.from_numpy(src)
    valid = (src_t >= 0)
    gather_idx = src_t.clamp_min(0).unsqueeze(-1).expand(B, Tm, x.shape[2])
    out = torch.gather(x, 1, gather_idx)
    return out * valid.unsqueeze(-1).to(x.dtype)


# --------------------------------------------------------------------------- #
# modules
# --------------------------------------------------------------------------- #
def series_predictor(x_idx: Tensor, P: Dict[str, Tensor], prefix: str, training: bool,
                     alpha: float = 1.0, new_buffers=None) -> Tensor:
    """SeriesPredictor.forward (forward_tacotron.py:28-39), dropout treated as p=0."""
    x = embedding(x_idx, P[prefix + 'embedding.weight']).transpose(1, 2)
    for i in range(3):
        x = batchnorm_conv(x, P, f'{prefix}convs.{i}.', True, training, new_buffers)
    x = x.transpose(1, 2)
    x = bigru(x, P, prefix + 'rnn.')
    x = linear(x, P[prefix + 'lin.weight'], P[prefix + 'lin.bias'])
    return x / alpha


def cbhg(x: Tensor, P: Dict[str, Tensor], prefix: str, K: int, num_highways: int,
         training: bool, new_buffers=None) -> Tensor:
    """CBHG.forward (common_layers.py:91-124), dropout treated as p=0.  x [B,C,T] -> [B,T,2*channels]."""
    residual = x
    T = x.shape[-1]
    bank = []
    for k in range(1, K + 1):
        c = batchnorm_conv(x, P, f'{prefix}conv1d_bank.{k - 1}.', True, training, new_buffers)
        bank.append(c[:, :, :T])
    y = torch.cat(bank, dim=1)
    y = maxpool_k2s1p1(y)
    y = batchnorm_conv(y, P, prefix + 'conv_project1.', True, training, new_buffers)
    y = batchnorm_conv(y, P, prefix + 'conv_project2.', False, training, new_buffers)
    y = y + residual
    y = y.transpose(1, 2)
    y = linear(y, P[prefix + 'pre_highway.weight'])
    for i in range(num_highways):
        y = highway(y, P, f'{prefix}highways.{i}.')
    return bigru(y, P, prefix + 'rnn.')


def pad_to(x: Tensor, max_len: int, pad_value: float = PAD_VALUE) -> Tensor:
    """ForwardTacotron._pad (forward_tacotron.py:236-239)."""
    x = x[:, :, :max_len]
    if x.shape[2] < max_len:
        fill = torch.full((x.shape[0], x.shape[1], max_len - x.shape[2]), pad_value, dtype=x.dtype)
        x = torch.cat([x, fill], dim=2)
    return x


def _trunk(x_idx, dur, pitch, energy, mel_lens, P, cfg, training, new_buffers, pad_frames_to=None):
    """pad_frames_to: zero-pad the LengthRegulator output to this many frames -- what pad_sequence does to an item that
    sits in a batch beside a longer one (common_layers.py:23); lets one item of a large batch be checked alone."""
    x = embedding(x_idx, P['embedding.weight']).transpose(1, 2)
    x = cbhg(x, P, 'prenet.', cfg['prenet_k'], cfg['prenet_num_highways'], training, new_buffers)
    pp = conv1d(pitch, P['pitch_proj.weight'], P['pitch_proj.bias']).transpose(1, 2)
    x = x + pp * cfg['pitch_strength']
    ep = conv1d(energy, P['energy_proj.weight'], P['energy_proj.bias']).transpose(1, 2)
    x = x + ep * cfg['energy_strength']
    x = length_regulate(x, dur)
    if pad_frames_to is not None and x.shape[1] < pad_frames_to:
        x = torch.cat([x, torch.zeros(x.shape[0], pad_frames_to - x.shape[1], x.shape[2], dtype=x.dtype)], dim=1)
    x = bilstm(x, mel_lens, P, 'lstm.', cfg.get('padding_value', PAD_VALUE))
    x = linear(x, P['lin.weight'], P['lin.bias']).transpose(1, 2)
    xp = cbhg(x, P, 'postnet.', cfg['postnet_k'], cfg['postnet_num_highways'], training, new_buffers)
    xp = linear(xp, P['post_proj.weight']).transpose(1, 2)
    return x, xp


def forward(P: Dict[str, Tensor], batch: Dict[str, Tensor], cfg: dict, training: bool
            ) -> Tuple[Dict[str, Tensor], Dict[str, Tensor]]:
    """ForwardTacotron.forward (forward_tacotron.py:118-165) with all dropout = 0.

    Returns (outputs, new_buffers).  new_buffers holds the BN running stats /
    num_batches_tracked / step values after the call (training mode only).
    Mutates batch['dur'] in place like the reference (common_layers.py:18).
    """
    new_buffers: Dict[str, Tensor] = {}
    x_idx, mel, dur, mel_lens = batch['x'], batch['mel'], batch['dur'], batch['mel_len']
    pitch = batch['pitch'].unsqueeze(1)
    energy = batch['energy'].unsqueeze(1)
    if training:
        new_buffers['step'] = P['step'] + 1
    dur_hat = series_predictor(x_idx, P, 'dur_pred.', training, 1.0, new_buffers).squeeze(-1)
    pitch_hat = series_predictor(x_idx, P, 'pitch_pred.', training, 1.0, new_buffers).transpose(1, 2)
    energy_hat = series_predictor(x_idx, P, 'energy_pred.', training, 1.0, new_buffers).transpose(1, 2)
    x, xp = _trunk(x_idx, dur, pitch, energy, mel_lens, P, cfg, training, new_buffers)
    pv = cfg.get('padding_value', PAD_VALUE)
    out = {'mel': pad_to(x, mel.shape[2], pv), 'mel_post': pad_to(xp, mel.shape[2], pv),
           'dur': dur_hat, 'pitch': pitch_hat, 'energy': energy_hat}
    return out, new_buffers


def generate(P: Dict[str, Tensor], x_idx: Tensor, cfg: dict, alpha: float = 1.0,
             pitch_function=lambda x: x, energy_function=lambda x: x) -> Dict[str, Tensor]:
    """ForwardTacotron.generate + _generate_mel (forward_tacotron.py:167-184, 205-234)."""
    with torch.no_grad():
        dur_hat = series_predictor(x_idx, P, 'dur_pred.', False, alpha).squeeze(2)
        if torch.sum(dur_hat.long()) <= 0:
            dur_hat = torch.full_like(dur_hat, 2.)
        pitch_hat = pitch_function(series_predictor(x_idx, P, 'pitch_pred.', False).transpose(1, 2))
        energy_hat = energy_function(series_predictor(x_idx, P, 'energy_pred.', False).transpose(1, 2))
        x, xp = _trunk(x_idx, dur_hat, pitch_hat, energy_hat, None, P, cfg, False, None)
        return {'mel': x, 'mel_post': xp, 'dur': dur_hat, 'pitch': pitch_hat, 'energy': energy_hat}


def generate_mel(P: Dict[str, Tensor], x_idx: Tensor, dur_hat: Tensor, pitch_hat: Tensor, energy_hat: Tensor,
                 cfg: dict, pad_frames_to: Optional[int] = None) -> Dict[str, Tensor]:
    """ForwardTacotron._generate_mel (forward_tacotron.py:205-234): eval-mode mel generation from GIVEN durations
    [B,Tx], pitch / energy [B,1,Tx].  pad_frames_to: see _trunk."""
    with torch.no_grad():
        x, xp = _trunk(x_idx, dur_hat, pitch_hat, energy_hat, None, P, cfg, False, None, pad_frames_to)
        return {'mel': x, 'mel_post': xp, 'dur': dur_hat, 'pitch': pitch_hat, 'energy': energy_hat}


# --------------------------------------------------------------------------- #
# trainer step  (trainer/forward_trainer.py:73-99, trainer/common.py:69-92)
# --------------------------------------------------------------------------- #
def pad_mask(lens: Tensor, max_len: int, dtype) -> Tensor:
    return (torch.arange(max_len).unsqueeze(0) < lens.unsqueeze(1)).to(dtype)


def masked_l1(x: Tensor, target: Tensor, lens: Tensor) -> Tensor:
    """MaskedL1.forward (trainer/common.py:71-78): sum|x*m - t*m| / sum(m expanded over channels)."""
    mask = pad_mask(lens, target.shape[2], x.dtype).unsqueeze(1).expand_as(x)
    return (x * mask - target * mask).abs().sum() / mask.sum()


def losses(pred: Dict[str, Tensor], batch: Dict[str, Tensor], pitch_target: Tensor,
           energy_target: Tensor, train_cfg: dict) -> Dict[str, Tensor]:
    """forward_trainer.py:83-93."""
    m1 = masked_l1(pred['mel'], batch['mel'], batch['mel_len'])
    m2 = masked_l1(pred['mel_post'], batch['mel'], batch['mel_len'])
    d = masked_l1(pred['dur'].unsqueeze(1), batch['dur'].unsqueeze(1), batch['x_len'])
    p = masked_l1(pred['pitch'], pitch_target.unsqueeze(1), batch['x_len'])
    e = masked_l1(pred['energy'], energy_target.unsqueeze(1), batch['x_len'])
    total = m1 + m2 + train_cfg['dur_loss_factor'] * d + train_cfg['pitch_loss_factor'] * p \
        + train_cfg['energy_loss_factor'] * e
    return {'loss': total, 'mel': m1, 'mel_post': m2, 'dur': d, 'pitch': p, 'energy': e}


def clip_grad_norm(grads: Dict[str, Tensor], max_norm: float) -> Tuple[Dict[str, Tensor], Tensor]:
    """torch.nn.utils.clip_grad_norm_ (forward_trainer.py:97-98): global L2 norm,
    scale by max_norm/(norm+1e-6) clamped to 1."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).to(
        next(iter(grads.values())).dtype)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return {k: g * coef for k, g in grads.items()}, total


def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
              b1=0.9, b2=0.999, eps=1e-8) -> Tuple[Tensor, Tensor, Tensor]:
    """torch.optim.Adam defaults (train_forward.py:76), no weight decay, no amsgrad."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * m / denom, m, v


PARAM_SUFFIXES = ('weight', 'bias', 'weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0',
                  'weight_ih_l0_reverse', 'weight_hh_l0_reverse', 'bias_ih_l0_reverse',
                  'bias_hh_l0_reverse')


def is_param(key: str) -> bool:
    """state_dict entries that are nn.Parameters (everything except BN buffers and 'step')."""
    last = key.split('.')[-1]
    return last in PARAM_SUFFIXES


def train_step(P: Dict[str, Tensor], opt_state: Dict[str, Dict[str, Tensor]], batch: Dict[str, Tensor],
               cfg: dict, train_cfg: dict, lr: float, step_count: int):
    """One full optimisation step (forward_trainer.py:73-99) with dropout/zoneout = 0.

    Returns (new_P, new_opt_state, info) where info has losses, grads (pre-clip),
    grad_norm and the forward outputs.
    """
    leaf = {k: (v.detach().clone().requires_grad_(True) if is_param(k) else v) for k, v in P.items()}
    b = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
    pitch_target = b['pitch'].detach().clone()
    energy_target = b['energy'].detach().clone()
    pred, new_buf = forward(leaf, b, cfg, training=True)
    L = losses(pred, b, pitch_target, energy_target, train_cfg)
    names = [k for k in leaf if is_param(k)]
    gl = torch.autograd.grad(L['loss'], [leaf[k] for k in names], allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(leaf[k])) for k, g in zip(names, gl)}
    clipped, gnorm = clip_grad_norm(grads, train_cfg['clip_grad_norm'])
    new_P = dict(P)
    new_opt = {}
    for k in names:
        st = opt_state.get(k) or {'m': torch.zeros_like(P[k]), 'v': torch.zeros_like(P[k])}
        p, m, v = adam_step(P[k], clipped[k], st['m'], st['v'], step_count, lr)
        new_P[k] = p
        new_opt[k] = {'m': m, 'v': v}
    for k, v in new_buf.items():
        new_P[k] = v
    info = {'losses': {k: v.detach() for k, v in L.items()}, 'grads': grads, 'grad_norm': gnorm,
            'pred': {k: v.detach() for k, v in pred.items()}}
    return new_P, new_opt, info


# --------------------------------------------------------------------------- #
# multispeaker variant  (models/multi_forward_tacotron.py, trainer/multi_forward_trainer.py)
# --------------------------------------------------------------------------- #
def _speaker_cat(parts: List[Tensor], semb: Tensor) -> Tensor:
    """torch.cat([..., semb[:, None, :].repeat(1, T, 1)], dim=2)  (multi_forward_tacotron.py:39-42,83-85)."""
    T = parts[0].shape[1]
    return torch.cat(parts + [semb[:, None, :].repeat(1, T, 1)], dim=2)


def multi_series_predictor(x_idx, semb, P, prefix, training, alpha=1.0, new_buffers=None, x_cond=None):
    """SeriesPredictor / ConditionalSeriesPredictor.forward (multi_forward_tacotron.py:35-50, 76-93)."""
    parts = [embedding(x_idx, P[prefix + 'embedding.weight'])]
    if x_cond is not None:
        parts.append(embedding(x_cond, P[prefix + 'pitch_cond_embedding.weight']))
    x = _speaker_cat(parts, semb).transpose(1, 2)
    for i in range(3):
        x = batchnorm_conv(x, P, f'{prefix}convs.{i}.', True, training, new_buffers)
    x = bigru(x.transpose(1, 2), P, prefix + 'rnn.')
    return linear(x, P[prefix + 'lin.weight'], P[prefix + 'lin.bias']) / alpha


def multi_forward(P, batch, cfg, training):
    """MultiForwardTacotron.forward (multi_forward_tacotron.py:186-241), dropout = 0."""
    nb: Dict[str, Tensor] = {}
    x_idx, mel, dur, semb, mel_lens = batch['x'], batch['mel'], batch['dur'], batch['speaker_emb'], batch['mel_len']
    pitch, energy, pitch_cond = batch['pitch'].unsqueeze(1), batch['energy'].unsqueeze(1), batch['pitch_cond']
    if training:
        nb['step'] = P['step'] + 1
    pc_hat = multi_series_predictor(x_idx, semb, P, 'pitch_cond_pred.', training, 1.0, nb)
    dur_hat = multi_series_predictor(x_idx, semb, P, 'dur_pred.', training, 1.0, nb, pitch_cond).squeeze(-1)
    pitch_hat = multi_series_predictor(x_idx, semb, P, 'pitch_pred.', training, 1.0, nb, pitch_cond).transpose(1, 2)
    energy_hat = multi_series_predictor(x_idx, semb, P, 'energy_pred.', training, 1.0, nb).transpose(1, 2)
    x = embedding(x_idx, P['embedding.weight']).transpose(1, 2)
    x = cbhg(x, P, 'prenet.', cfg['prenet_k'], cfg['prenet_num_highways'], training, nb)
    x = _speaker_cat([x], semb)
    x = x + conv1d(pitch, P['pitch_proj.weight'], P['pitch_proj.bias']).transpose(1, 2) * cfg['pitch_strength']
    x = x + conv1d(energy, P['energy_proj.weight'], P['energy_proj.bias']).transpose(1, 2) * cfg['energy_strength']
    x = length_regulate(x, dur)
    pv = cfg.get('padding_value', PAD_VALUE)
    x = bilstm(x, mel_lens, P, 'lstm.', pv)
    x = linear(x, P['lin.weight'], P['lin.bias']).transpose(1, 2)
    xp = cbhg(x, P, 'postnet.', cfg['postnet_k'], cfg['postnet_num_highways'], training, nb)
    xp = linear(xp, P['post_proj.weight']).transpose(1, 2)
    out = {'mel': pad_to(x, mel.shape[2], pv), 'mel_post': pad_to(xp, mel.shape[2], pv), 'dur': dur_hat,
           'pitch': pitch_hat, 'energy': energy_hat, 'pitch_cond': pc_hat}
    return out, nb


def cross_entropy(logits: Tensor, target: Tensor, ignore_index: int = 0) -> Tensor:
    """nn.CrossEntropyLoss(ignore_index=0) (multi_forward_trainer.py:34,88) on logits [...,K], target [...]."""
    K = logits.shape[-1]
    l2 = logits.reshape(-1, K)
    t = target.reshape(-1)
    keep = t != ignore_index
    lse = torch.logsumexp(l2, dim=1)
    picked = l2.gather(1, t.clamp(0, K - 1).unsqueeze(1)).squeeze(1)
    return ((lse - picked) * keep.to(l2.dtype)).sum() / keep.sum()


def multi_train_step(P, opt_state, batch, cfg, train_cfg, lr, step_count):
    """multi_forward_trainer.py:73-99 with dropout = 0."""
    leaf = {k: (v.detach().clone().requires_grad_(True) if is_param(k) else v) for k, v in P.items()}
    b = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
    pitch_target = b['pitch'].detach().clone()
    energy_target = b['energy'].detach().clone()
    pred, new_buf = multi_forward(leaf, b, cfg, training=True)
    L = losses(pred, b, pitch_target, energy_target, train_cfg)
    ce = cross_entropy(pred['pitch_cond'], b['pitch_cond'], 0)
    L['pitch_cond'] = ce
    L['loss'] = L['loss'] + train_cfg['pitch_cond_loss_factor'] * ce
    names = [k for k in leaf if is_param(k)]
    gl = torch.autograd.grad(L['loss'], [leaf[k] for k in names], allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(leaf[k])) for k, g in zip(names, gl)}
    clipped, gnorm = clip_grad_norm(grads, train_cfg['clip_grad_norm'])
    new_P = dict(P)
    new_opt = {}
    for k in names:
        st = opt_state.get(k) or {'m': torch.zeros_like(P[k]), 'v': torch.zeros_like(P[k])}
        p, m, v = adam_step(P[k], clipped[k], st['m'], st['v'], step_count, lr)
        new_P[k] = p
        new_opt[k] = {'m': m, 'v': v}
    for k, v in new_buf.items():
        new_P[k] = v
    info = {'losses': {k: v.detach() for k, v in L.items()}, 'grads': grads, 'grad_norm': gnorm,
            'pred': {k: v.detach() for k, v in pred.items()}}
    return new_P, new_opt, info


# --------------------------------------------------------------------------- #
# synthetic LJSpeech-shaped batch   -- shared by tests and bench
# --------------------------------------------------------------------------- #
def synthetic_batch(B: int = 32, Tmax: int = 128, n_mels: int = 80, num_chars: int = 135,
                    max_dur: int = 12, seed: int = 0) -> Dict[str, Tensor]:
    g = torch.Generator().manual_seed(seed)
    x_len = torch.randint(Tmax // 2, Tmax + 1, (B,), generator=g)
    x_len[0] = Tmax
    x = torch.zeros(B, Tmax, dtype=torch.long)
    dur = torch.zeros(B, Tmax)
    for b in range(B):
        L = int(x_len[b])
        x[b, :L] = torch.randint(1, num_chars, (L,), generator=g)
        dur[b, :L] = torch.randint(1, max_dur, (L,), generator=g).float()
    mel_len = dur.sum(1).long()
    Tm = int(mel_len.max())
    mel = torch.full((B, n_mels, Tm + 1), PAD_VALUE)
    for b in range(B):
        n = int(mel_len[b])
        mel[b, :, :n] = torch.randn(n_mels, n, generator=g) * 2 - 5
    pitch = torch.randn(B, Tmax, generator=g) * (x > 0)
    energy = torch.rand(B, Tmax, generator=g) * (x > 0)
    return {'x': x, 'mel': mel, 'dur': dur, 'x_len': x_len, 'mel_len': mel_len,
            'pitch': pitch, 'energy': energy}
