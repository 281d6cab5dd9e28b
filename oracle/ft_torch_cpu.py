"""CPU BASELINE -- TEST / MEASUREMENT INFRASTRUCTURE ONLY.  Never imported by the product path.

The reference's train step restated on STOCK FUSED torch CPU operators -- `F.conv1d`, `F.batch_norm`,
`F.max_pool1d`, `F.linear`, `F.embedding`, `torch._VF.gru` / `torch._VF.lstm` (the packed-sequence form for the
trunk LSTM), `repeat_interleave` + `pad_sequence`, autograd, `clip_grad_norm_`, `torch.optim.Adam` -- i.e. the
operators the reference's nn.Modules dispatch to on a CPU, driven from a state_dict-keyed functional forward (no
nn.Module of the reference is instantiated or copied).  It is what `bench.py` times as `cpu_baseline` (SURVEY.md
section 8d / BASELINE.md section 4: "stock torch CPU ops, fp32, the identical seeded batch, 1 warm-up + >= 3 timed
steps, all host cores"), because the checker oracle (oracle/ft_oracle.py: per-timestep Python loops, tap-sum
convolutions) is ~10x slower than the thing the baseline stands for.

Pinning: tests/test_cpu_baseline.py checks this file against oracle/ft_oracle.py (which is pinned to the reference's
goldens) AND against the goldens directly: outputs, losses, every gradient, post-Adam parameters, BN statistics.

Reference lines followed: models/forward_tacotron.py:28-39 (SeriesPredictor), :118-165 (forward);
models/common_layers.py:17-24 (LengthRegulator), :35-40 (Highway), :54-57 (BatchNormConv), :91-124 (CBHG);
trainer/forward_trainer.py:73-99 (step), trainer/common.py:69-92 (MaskedL1).  Dropout / zoneout are 0 (parity runs)
or left out (timing: the reference's dropouts are negligible next to its recurrences).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence, pad_sequence

Tensor = torch.Tensor
PAD_VALUE = -11.5129
_RNN_KEYS = ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0',
             'weight_ih_l0_reverse', 'weight_hh_l0_reverse', 'bias_ih_l0_reverse', 'bias_hh_l0_reverse')


def _bnconv(x: Tensor, P: Dict[str, Tensor], pre: str, relu: bool, training: bool, stats: Optional[dict]) -> Tensor:
    w = P[pre + 'conv.weight']
    y = F.conv1d(x, w, None, 1, w.shape[2] // 2)
    if relu:
        y = F.relu(y)
    rm, rv = P[pre + 'bnorm.running_mean'], P[pre + 'bnorm.running_var']
    if training:            # F.batch_norm updates its running-stat arguments in place: give it copies
        rm, rv = rm.clone(), rv.clone()
    y = F.batch_norm(y, rm, rv, P[pre + 'bnorm.weight'], P[pre + 'bnorm.bias'], training, 0.1, 1e-5)
    if training and stats is not None:
        stats[pre + 'bnorm.running_mean'] = rm
        stats[pre + 'bnorm.running_var'] = rv
        stats[pre + 'bnorm.num_batches_tracked'] = P[pre + 'bnorm.num_batches_tracked'] + 1
    return y


def _rnn_weights(P: Dict[str, Tensor], pre: str) -> List[Tensor]:
    return [P[pre + k] for k in _RNN_KEYS]


def _bigru(x: Tensor, P: Dict[str, Tensor], pre: str, training: bool) -> Tensor:
    H = P[pre + 'weight_hh_l0'].shape[1]
    h0 = x.new_zeros(2, x.shape[0], H)
    return torch._VF.gru(x, h0, _rnn_weights(P, pre), True, 1, 0.0, training, True, True)[0]


def _bilstm_packed(x: Tensor, lens: Optional[Tensor], P: Dict[str, Tensor], pre: str, training: bool,
                   pad_value: float) -> Tensor:
    H = P[pre + 'weight_hh_l0'].shape[1]
    if lens is None:
        h0 = x.new_zeros(2, x.shape[0], H)
        return torch._VF.lstm(x, (h0, h0.clone()), _rnn_weights(P, pre), True, 1, 0.0, training, True, True)[0]
    pk = pack_padded_sequence(x, lens.cpu(), enforce_sorted=False, batch_first=True)
    nb = int(pk.batch_sizes[0])
    h0 = x.new_zeros(2, nb, H)
    out = torch._VF.lstm(pk.data, pk.batch_sizes, (h0, h0.clone()), _rnn_weights(P, pre), True, 1, 0.0, training,
                         True)[0]
    pk = torch.nn.utils.rnn.PackedSequence(out, pk.batch_sizes, pk.sorted_indices, pk.unsorted_indices)
    return pad_packed_sequence(pk, padding_value=pad_value, batch_first=True)[0]


def _predictor(x_idx: Tensor, P: Dict[str, Tensor], pre: str, training: bool, stats, alpha: float = 1.0) -> Tensor:
    x = F.embedding(x_idx, P[pre + 'embedding.weight']).transpose(1, 2)
    for i in range(3):
        x = _bnconv(x, P, f'{pre}convs.{i}.', True, training, stats)
    x = _bigru(x.transpose(1, 2), P, pre + 'rnn.', training)
    return F.linear(x, P[pre + 'lin.weight'], P[pre + 'lin.bias']) / alpha


def _cbhg(x: Tensor, P: Dict[str, Tensor], pre: str, K: int, n_hw: int, training: bool, stats) -> Tensor:
    T = x.shape[-1]
    bank = torch.cat([_bnconv(x, P, f'{pre}conv1d_bank.{k}.', True, training, stats)[:, :, :T] for k in range(K)], 1)
    y = F.max_pool1d(bank, 2, 1, 1)[:, :, :T]
    y = _bnconv(y, P, pre + 'conv_project1.', True, training, stats)
    y = _bnconv(y, P, pre + 'conv_project2.', False, training, stats)
    y = F.linear((y + x).transpose(1, 2), P[pre + 'pre_highway.weight'])
    for i in range(n_hw):
        hp = f'{pre}highways.{i}.'
        x1 = F.linear(y, P[hp + 'W1.weight'], P[hp + 'W1.bias'])
        g = torch.sigmoid(F.linear(y, P[hp + 'W2.weight'], P[hp + 'W2.bias']))
        y = g * F.relu(x1) + (1. - g) * y
    return _bigru(y, P, pre + 'rnn.', training)


def _length_regulate(x: Tensor, dur: Tensor) -> Tensor:
    dur[dur < 0] = 0.
    reps = (dur + 0.5).long()
    return pad_sequence([torch.repeat_interleave(x[b], reps[b], dim=0) for b in range(x.shape[0])],
                        batch_first=True, padding_value=0.)


def _pad(x: Tensor, n: int, value: float) -> Tensor:
    x = x[:, :, :n]
    return F.pad(x, [0, n - x.shape[2], 0, 0], 'constant', value)


def forward(P: Dict[str, Tensor], batch: Dict[str, Tensor], cfg: dict, training: bool
            ) -> Tuple[Dict[str, Tensor], Dict[str, Tensor]]:
    """-> (outputs, updated buffers); mutates batch['dur'] in place like the reference"""
    stats: Dict[str, Tensor] = {}
    if training:
        stats['step'] = P['step'] + 1
    x_idx, mel, dur, mel_lens = batch['x'], batch['mel'], batch['dur'], batch['mel_len']
    dur_hat = _predictor(x_idx, P, 'dur_pred.', training, stats).squeeze(-1)
    pitch_hat = _predictor(x_idx, P, 'pitch_pred.', training, stats).transpose(1, 2)
    energy_hat = _predictor(x_idx, P, 'energy_pred.', training, stats).transpose(1, 2)
    x = F.embedding(x_idx, P['embedding.weight']).transpose(1, 2)
    x = _cbhg(x, P, 'prenet.', cfg['prenet_k'], cfg['prenet_num_highways'], training, stats)
    pp = F.conv1d(batch['pitch'].unsqueeze(1), P['pitch_proj.weight'], P['pitch_proj.bias'], 1, 1)
    ep = F.conv1d(batch['energy'].unsqueeze(1), P['energy_proj.weight'], P['energy_proj.bias'], 1, 1)
    x = x + pp.transpose(1, 2) * cfg['pitch_strength'] + ep.transpose(1, 2) * cfg['energy_strength']
    x = _length_regulate(x, dur)
    pv = cfg.get('padding_value', PAD_VALUE)
    x = _bilstm_packed(x, mel_lens, P, 'lstm.', training, pv)
    x = F.linear(x, P['lin.weight'], P['lin.bias']).transpose(1, 2)
    xp = _cbhg(x, P, 'postnet.', cfg['postnet_k'], cfg['postnet_num_highways'], training, stats)
    xp = F.linear(xp, P['post_proj.weight']).transpose(1, 2)
    out = {'mel': _pad(x, mel.shape[2], pv), 'mel_post': _pad(xp, mel.shape[2], pv), 'dur': dur_hat,
           'pitch': pitch_hat, 'energy': energy_hat}
    return out, stats


def _masked_l1(x: Tensor, target: Tensor, lens: Tensor) -> Tensor:
    m = (torch.arange(target.shape[2]).unsqueeze(0) < lens.unsqueeze(1)).to(x.dtype).unsqueeze(1).expand_as(x)
    return F.l1_loss(x * m, target * m, reduction='sum') / m.sum()


def is_param(key: str) -> bool:
    return key.split('.')[-1] in ('weight', 'bias') + _RNN_KEYS


class CpuTrainer:
    """state_dict-keyed parameters as autograd leaves + torch.optim.Adam (train_forward.py:76)."""

    def __init__(self, P: Dict[str, Tensor], cfg: dict, train_cfg: dict, lr: float):
        self.P = {k: (v.detach().clone().requires_grad_(True) if is_param(k) else v.clone()) for k, v in P.items()}
        self.cfg, self.tc = cfg, train_cfg
        self.names = [k for k in self.P if is_param(k)]
        self.opt = torch.optim.Adam([self.P[k] for k in self.names], lr=lr)

    def step(self, batch: Dict[str, Tensor]) -> Dict[str, Tensor]:
        b = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        pt, et = b['pitch'].clone(), b['energy'].clone()
        pred, stats = forward(self.P, b, self.cfg, training=True)
        c = self.tc
        L = {'mel': _masked_l1(pred['mel'], b['mel'], b['mel_len']),
             'mel_post': _masked_l1(pred['mel_post'], b['mel'], b['mel_len']),
             'dur': _masked_l1(pred['dur'].unsqueeze(1), b['dur'].unsqueeze(1), b['x_len']),
             'pitch': _masked_l1(pred['pitch'], pt.unsqueeze(1), b['x_len']),
             'energy': _masked_l1(pred['energy'], et.unsqueeze(1), b['x_len'])}
        L['loss'] = L['mel'] + L['mel_post'] + c['dur_loss_factor'] * L['dur'] + c['pitch_loss_factor'] * L['pitch'] \
            + c['energy_loss_factor'] * L['energy']
        self.opt.zero_grad()
        L['loss'].backward()
        grads = {k: (self.P[k].grad.clone() if self.P[k].grad is not None else torch.zeros_like(self.P[k]))
                 for k in self.names}
        gn = torch.nn.utils.clip_grad_norm_([self.P[k] for k in self.names], c['clip_grad_norm'])
        self.opt.step()
        for k, v in stats.items():
            self.P[k] = v.detach()
        return {'losses': {k: v.detach() for k, v in L.items()}, 'grads': grads, 'grad_norm': gn.detach(),
                'pred': {k: v.detach() for k, v in pred.items()}}

    def state_dict(self) -> Dict[str, Tensor]:
        return {k: v.detach() for k, v in self.P.items()}
