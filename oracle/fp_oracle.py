"""CPU oracle for the FastPitch variant -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product path
(forwardtacotron_amd/) never does.  Restates, in explicit torch-CPU fp32 arithmetic, what the reference computes in
  models/fast_pitch.py:14-41 (SeriesPredictor), :123-165 (forward), :167-221 (generate / _generate_mel)
  models/common_layers.py:127-145 (PositionalEncoding), :148-185 (FFTBlock), :188-223 (ForwardTransformer)
with torch.nn.MultiheadAttention written out (torch/nn/functional.py multi_head_attention_forward, the
need_weights=True branch the reference takes: q scaled by 1/sqrt(hd), additive -inf key padding mask, softmax,
P @ V, output projection).  Pinned by tests/golden/tiny_fastpitch.npz, produced by importing the reference
(tests/golden/make_golden.py fastpitch).  All dropout = 0.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch

from .ft_oracle import (PAD_VALUE, adam_step, clip_grad_norm, conv1d, embedding, length_regulate, linear, losses,
                        pad_to)

Tensor = torch.Tensor


def layernorm(x: Tensor, g: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def mha(x: Tensor, key_pad: Optional[Tensor], P: Dict[str, Tensor], prefix: str, nheads: int) -> Tensor:
    """self-attention on batch-major x [B,T,d]; key_pad bool [B,T] (True = ignore that key)."""
    B, T, d = x.shape
    hd = d // nheads
    qkv = linear(x, P[prefix + 'in_proj_weight'], P[prefix + 'in_proj_bias'])
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]

    def heads(t):
        return t.reshape(B, T, nheads, hd).permute(0, 2, 1, 3)          # [B,nh,T,hd]

    q, k, v = heads(q) * math.sqrt(1.0 / hd), heads(k), heads(v)
    s = q @ k.transpose(-1, -2)                                          # [B,nh,T,T]
    if key_pad is not None:
        s = s.masked_fill(key_pad[:, None, None, :], float('-inf'))
    p = torch.softmax(s, dim=-1)
    a = (p @ v).permute(0, 2, 1, 3).reshape(B, T, d)
    return linear(a, P[prefix + 'out_proj.weight'], P[prefix + 'out_proj.bias'])


def fft_block(x: Tensor, key_pad: Optional[Tensor], P: Dict[str, Tensor], prefix: str, nheads: int) -> Tensor:
    """common_layers.py:170-185"""
    x = layernorm(x + mha(x, key_pad, P, prefix + 'self_attn.', nheads), P[prefix + 'norm1.weight'],
                  P[prefix + 'norm1.bias'])
    h = conv1d(x.transpose(1, 2), P[prefix + 'conv1.weight'], P[prefix + 'conv1.bias'])
    h = torch.relu(h)
    h = conv1d(h, P[prefix + 'conv2.weight'], P[prefix + 'conv2.bias']).transpose(1, 2)
    return layernorm(x + h, P[prefix + 'norm2.weight'], P[prefix + 'norm2.bias'])


def forward_transformer(x: Tensor, key_pad: Optional[Tensor], P: Dict[str, Tensor], prefix: str, nheads: int,
                        layers: int) -> Tensor:
    """common_layers.py:214-223 (batch-major throughout; pe [max_len,1,d])"""
    T = x.shape[1]
    x = x + P[prefix + 'pos_encoder.scale'] * P[prefix + 'pos_encoder.pe'][:T, 0, :].unsqueeze(0)
    for i in range(layers):
        x = fft_block(x, key_pad, P, f'{prefix}layers.{i}.', nheads)
    return layernorm(x, P[prefix + 'norm.weight'], P[prefix + 'norm.bias'])


def series_predictor(x_idx: Tensor, key_pad: Optional[Tensor], P, prefix: str, nheads: int, layers: int,
                     alpha: float = 1.0) -> Tensor:
    """fast_pitch.py:33-41"""
    x = embedding(x_idx, P[prefix + 'embedding.weight'])
    x = forward_transformer(x, key_pad, P, prefix + 'transformer.', nheads, layers)
    return linear(x, P[prefix + 'lin.weight'], P[prefix + 'lin.bias']) / alpha


def _mel(x_idx, tok_mask, dur, pitch, energy, frame_mask_lens, P, cfg) -> Tensor:
    x = embedding(x_idx, P['embedding.weight'])
    x = forward_transformer(x, tok_mask, P, 'prenet.', cfg['prenet_heads'], cfg['prenet_layers'])
    x = x + conv1d(pitch, P['pitch_proj.weight'], P['pitch_proj.bias']).transpose(1, 2) * cfg['pitch_strength']
    x = x + conv1d(energy, P['energy_proj.weight'], P['energy_proj.bias']).transpose(1, 2) * cfg['energy_strength']
    x = length_regulate(x, dur)
    fm = None
    if frame_mask_lens is not None:
        fm = torch.arange(x.shape[1]).unsqueeze(0) >= frame_mask_lens.unsqueeze(1)
    x = forward_transformer(x, fm, P, 'postnet.', cfg['postnet_heads'], cfg['postnet_layers'])
    return linear(x, P['lin.weight'], P['lin.bias']).transpose(1, 2)


def forward(P: Dict[str, Tensor], batch: Dict[str, Tensor], cfg: dict, training: bool
            ) -> Tuple[Dict[str, Tensor], Dict[str, Tensor]]:
    """FastPitch.forward (fast_pitch.py:123-165); mutates batch['dur'] like the reference's LengthRegulator."""
    new_buffers: Dict[str, Tensor] = {}
    x_idx, mel, dur, mel_lens = batch['x'], batch['mel'], batch['dur'], batch['mel_len']
    if training:
        new_buffers['step'] = P['step'] + 1
    m = x_idx == 0
    dur_hat = series_predictor(x_idx, m, P, 'dur_pred.', cfg['durpred_n_heads'], cfg['durpred_layers']).squeeze(-1)
    pitch_hat = series_predictor(x_idx, m, P, 'pitch_pred.', cfg['pitch_n_heads'], cfg['pitch_layers']).transpose(1, 2)
    energy_hat = series_predictor(x_idx, m, P, 'energy_pred.', cfg['energy_n_heads'],
                                  cfg['energy_layers']).transpose(1, 2)
    x = _mel(x_idx, m, dur, batch['pitch'].unsqueeze(1), batch['energy'].unsqueeze(1), mel_lens, P, cfg)
    pv = cfg.get('padding_value', PAD_VALUE)
    x = pad_to(x, mel.shape[2], pv)
    return {'mel': x, 'mel_post': x, 'dur': dur_hat, 'pitch': pitch_hat, 'energy': energy_hat}, new_buffers


def generate(P: Dict[str, Tensor], x_idx: Tensor, cfg: dict, alpha: float = 1.0,
             pitch_function=lambda x: x, energy_function=lambda x: x) -> Dict[str, Tensor]:
    """fast_pitch.py:167-221: predictors run unmasked, prenet masked by x==0, postnet unmasked."""
    with torch.no_grad():
        dur_hat = series_predictor(x_idx, None, P, 'dur_pred.', cfg['durpred_n_heads'], cfg['durpred_layers'],
                                   alpha).squeeze(2)
        if torch.sum(dur_hat.long()) <= 0:
            dur_hat = torch.full_like(dur_hat, 2.)
        pitch_hat = pitch_function(series_predictor(x_idx, None, P, 'pitch_pred.', cfg['pitch_n_heads'],
                                                    cfg['pitch_layers']).transpose(1, 2))
        energy_hat = energy_function(series_predictor(x_idx, None, P, 'energy_pred.', cfg['energy_n_heads'],
                                                      cfg['energy_layers']).transpose(1, 2))
        x = _mel(x_idx, x_idx == 0, dur_hat, pitch_hat, energy_hat, None, P, cfg)
        return {'mel': x, 'mel_post': x, 'dur': dur_hat, 'pitch': pitch_hat, 'energy': energy_hat}


NON_PARAMS = ('pe', 'step')


def is_param(key: str) -> bool:
    return key.split('.')[-1] not in NON_PARAMS


def train_step(P, opt_state, batch, cfg, train_cfg, lr: float, step_count: int):
    """One optimisation step of the reference's ForwardTrainer (trainer/forward_trainer.py:73-99) on FastPitch."""
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
# MultiFastPitch  (models/multi_fast_pitch.py) -- pinned by tests/golden/tiny_multi_fastpitch.npz
# --------------------------------------------------------------------------- #
def _cat_speaker(parts, semb: Tensor) -> Tensor:
    T = parts[0].shape[1]
    return torch.cat(list(parts) + [semb[:, None, :].repeat(1, T, 1)], dim=2)


def multi_series_predictor(x_idx, semb, key_pad, P, prefix, nheads, layers, alpha=1.0, x_cond=None) -> Tensor:
    """multi_fast_pitch.py:36-49 (x_cond None) / :76-90 (conditional)"""
    parts = [embedding(x_idx, P[prefix + 'embedding.weight'])]
    if x_cond is not None:
        parts.append(embedding(x_cond, P[prefix + 'conditional_embedding.weight']))
    x = _cat_speaker(parts, semb)
    x = forward_transformer(x, key_pad, P, prefix + 'transformer.', nheads, layers)
    return linear(x, P[prefix + 'lin.weight'], P[prefix + 'lin.bias']) / alpha


def _multi_mel(x_idx, semb, tok_mask, dur, pitch, energy, frame_lens, P, cfg) -> Tensor:
    x = _cat_speaker([embedding(x_idx, P['embedding.weight'])], semb)
    x = forward_transformer(x, tok_mask, P, 'prenet.', cfg['prenet_heads'], cfg['prenet_layers'])
    x = x + conv1d(pitch, P['pitch_proj.weight'], P['pitch_proj.bias']).transpose(1, 2) * cfg['pitch_strength']
    x = x + conv1d(energy, P['energy_proj.weight'], P['energy_proj.bias']).transpose(1, 2) * cfg['energy_strength']
    x = length_regulate(x, dur)
    fm = None
    if frame_lens is not None:
        fm = torch.arange(x.shape[1]).unsqueeze(0) >= frame_lens.unsqueeze(1)
    x = forward_transformer(x, fm, P, 'postnet.', cfg['postnet_heads'], cfg['postnet_layers'])
    return linear(x, P['lin.weight'], P['lin.bias']).transpose(1, 2)


def multi_forward(P, batch, cfg, training: bool):
    """MultiFastPitch.forward (multi_fast_pitch.py:192-245)"""
    new_buffers: Dict[str, Tensor] = {}
    x_idx, mel, dur, semb = batch['x'], batch['mel'], batch['dur'], batch['speaker_emb']
    pc = batch['pitch_cond']
    if training:
        new_buffers['step'] = P['step'] + 1
    m = x_idx == 0
    dur_hat = multi_series_predictor(x_idx, semb, m, P, 'dur_pred.', cfg['durpred_n_heads'], cfg['durpred_layers'],
                                     x_cond=pc).squeeze(-1)
    pitch_hat = multi_series_predictor(x_idx, semb, m, P, 'pitch_pred.', cfg['pitch_n_heads'], cfg['pitch_layers'],
                                       x_cond=pc).transpose(1, 2)
    pc_hat = multi_series_predictor(x_idx, semb, m, P, 'pitch_cond_pred.', cfg['pitch_cond_n_heads'],
                                    cfg['pitch_cond_layers'])
    energy_hat = multi_series_predictor(x_idx, semb, m, P, 'energy_pred.', cfg['energy_n_heads'],
                                        cfg['energy_layers']).transpose(1, 2)
    x = _multi_mel(x_idx, semb, m, dur, batch['pitch'].unsqueeze(1), batch['energy'].unsqueeze(1), batch['mel_len'], P,
                   cfg)
    x = pad_to(x, mel.shape[2], cfg.get('padding_value', PAD_VALUE))
    return {'mel': x, 'mel_post': x, 'pitch_cond': pc_hat, 'dur': dur_hat, 'pitch': pitch_hat,
            'energy': energy_hat}, new_buffers


def multi_generate(P, x_idx, semb, cfg, alpha: float = 1.0) -> Dict[str, Tensor]:
    """MultiFastPitch.generate (multi_fast_pitch.py:247-313), B = 1 like the reference"""
    with torch.no_grad():
        pc = multi_series_predictor(x_idx, semb, None, P, 'pitch_cond_pred.', cfg['pitch_cond_n_heads'],
                                    cfg['pitch_cond_layers'], alpha).squeeze(-1)
        pc = torch.argmax(pc.squeeze(), dim=1).long().unsqueeze(0)
        dur_hat = multi_series_predictor(x_idx, semb, None, P, 'dur_pred.', cfg['durpred_n_heads'],
                                         cfg['durpred_layers'], alpha, x_cond=pc).squeeze(2)
        if torch.sum(dur_hat.long()) <= 0:
            dur_hat = torch.full_like(dur_hat, 2.)
        pitch_hat = multi_series_predictor(x_idx, semb, None, P, 'pitch_pred.', cfg['pitch_n_heads'],
                                           cfg['pitch_layers'], x_cond=pc).transpose(1, 2)
        energy_hat = multi_series_predictor(x_idx, semb, None, P, 'energy_pred.', cfg['energy_n_heads'],
                                            cfg['energy_layers']).transpose(1, 2)
        x = _multi_mel(x_idx, semb, x_idx == 0, dur_hat, pitch_hat, energy_hat, None, P, cfg)
        return {'mel': x, 'mel_post': x, 'dur': dur_hat, 'pitch_cond': pc, 'pitch': pitch_hat, 'energy': energy_hat}


def multi_train_step(P, opt_state, batch, cfg, train_cfg, lr: float, step_count: int):
    """trainer/multi_forward_trainer.py:73-99 on MultiFastPitch (CrossEntropy(ignore_index=0) on pitch_cond)."""
    from .ft_oracle import cross_entropy
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
