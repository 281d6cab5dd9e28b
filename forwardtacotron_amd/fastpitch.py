"""MI355X-native FastPitch: drop-in for models/fast_pitch.py:14-235 and the transformer layers of
models/common_layers.py:127-223 (same constructor kwargs, batch-dict forward()/generate(), state_dict incl. the
`pe` buffers).  torch.nn.{MultiheadAttention,LayerNorm,Conv1d,Linear,Embedding} objects are parameter
containers only; every computation is a gfx950 kernel behind the C ABI:
  attention = in_proj GEMM -> strided-batch QK^T (f32 MFMA) -> masked softmax -> strided-batch PV -> out_proj,
  backward = the five transposed products (dV, dP, dQ, dK via batched NT/NN/TN GEMMs) + softmax gradient.
Activations are batch-major channels-last [B,T,d] (the reference's [T,B,d] is only a view convention).
"""
import copy
import functools
import math
import os
from pathlib import Path
from typing import Any, Callable, Dict, Optional, Union

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib
from . import hip as H
from . import ops
from .model import LengthRegulator, NUM_CHARS_DEFAULT, PAD_VALUE, _dropout, _seed
from .ops import _c, _emit, _emit_multi

_F4 = 4


def _p(t):
    return None if t is None else t.data_ptr()


def _bgemm(kind, A_ptr, lda, sA0, sA1, B_ptr, ldb, sB0, sB1, C_ptr, ldc, sC0, sC1, M, N, K, nb0, nb1, device,
           padded: bool = False):
    """padded: the [T,T]-shaped operand is stored with its row stride rounded up to 4 (see MHAFn)"""
    if kind == 'tn':
        nbytes = _lib.query('ft_bgemm_tn_workspace', M, N, K, nb0, nb1)
        ws = H.workspace(nbytes, device)
        _lib.call('ft_bgemm_tn', A_ptr, lda, sA0, sA1, B_ptr, ldb, sB0, sB1, C_ptr, ldc, sC0, sC1, M, N, K, nb0, nb1,
                  int(padded), ws.data_ptr(), ws.numel(), H._stream())
    elif kind == 'nn':
        _lib.call('ft_bgemm_nn', A_ptr, lda, sA0, sA1, B_ptr, ldb, sB0, sB1, C_ptr, ldc, sC0, sC1, M, N, K, nb0, nb1,
                  int(padded), H._stream())
    else:
        _lib.call('ft_bgemm_nt', A_ptr, lda, sA0, sA1, B_ptr, ldb, sB0, sB1, C_ptr, ldc, sC0, sC1, M, N, K, nb0, nb1,
                  H._stream())


def precision_scoped(fn):
    """runs a model method under the model's `matmul_dtype` ('fp32' default, 'bf16' = BASELINE configs[2])"""
    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        with H.gemm_precision(getattr(self, 'matmul_dtype', 'fp32')):
            return fn(self, *a, **k)
    return wrapped


class MHAFn(Function):
    """nn.MultiheadAttention(d, nheads, dropout)(x, x, x, key_padding_mask=key_pad)[0]  (common_layers.py:172-174)
    on batch-major x [B,T,d]; key_pad uint8 [B,T] (1 = padded key) or None."""

    @staticmethod
    def forward(ctx, x, key_pad, in_w, in_b, out_w, out_b, nheads, p_drop, seed):
        x = _c(x)
        B, T, d = x.shape
        nh = int(nheads)
        hd = d // nh
        scale = 1.0 / math.sqrt(hd)
        qkv = H.linear_fwd(x, in_w, in_b)                                   # [B,T,3d]
        # bf16 mode: ONE flash-style kernel between the two projections -- no [B,h,T,T] tensor in memory, the backward
        # recomputes the probabilities (csrc/ft_attn.hip); FT_ATTN_FUSED=0: the five-launch form below in bf16 too
        if H.gemm_precision_mode() == 'bf16' and hd in (64, 128) and os.environ.get('FT_ATTN_FUSED', '1') == '1':
            att, lse2 = H.attn_fwd(qkv, key_pad, nh, scale, p_drop, seed)
            out = H.linear_fwd(att, out_w, out_b)
            ctx.save_for_backward(x, qkv, lse2, att, key_pad, in_w, in_b, out_w, out_b, lse2)
            ctx.meta = (nh, hd, scale, float(p_drop), int(seed))
            ctx.fused = True
            return out
        ctx.fused = False
        # the [T,T] score / probability matrices are kept with their row stride rounded up to 4 floats (pad columns
        # are zeros): T = 841 frames would otherwise push four of the six attention GEMMs off the 16-B-load paths
        Tp = (T + 3) // 4 * 4
        P = torch.empty(B, nh, T, Tp, device=x.device, dtype=x.dtype)
        q0 = qkv.data_ptr()
        _bgemm('nt', q0, 3 * d, T * 3 * d, hd, q0 + d * _F4, 3 * d, T * 3 * d, hd, P.data_ptr(), Tp, nh * T * Tp, T * Tp,
               T, T, hd, B, nh, x.device)
        # softmax and nn.MultiheadAttention's attention dropout in one pass; both P and dropout(P) are kept for backward
        Pd = torch.empty_like(P) if p_drop > 0 else P
        _lib.call('ft_softmax_fwd', P.data_ptr(), _p(key_pad), B, nh, T, T, Tp, scale,
                  Pd.data_ptr() if p_drop > 0 else None, float(p_drop), int(seed), H._stream())
        att = torch.empty(B, T, d, device=x.device, dtype=x.dtype)
        _bgemm('nn', Pd.data_ptr(), Tp, nh * T * Tp, T * Tp, q0 + 2 * d * _F4, 3 * d, T * 3 * d, hd, att.data_ptr(), d,
               T * d, hd, T, hd, T, B, nh, x.device, padded=True)
        out = H.linear_fwd(att, out_w, out_b)
        ctx.save_for_backward(x, qkv, P, att, key_pad, in_w, in_b, out_w, out_b, Pd)
        ctx.meta = (nh, hd, scale, float(p_drop), int(seed))
        return out

    @staticmethod
    def backward(ctx, dout):
        x, qkv, P, att, key_pad, in_w, in_b, out_w, out_b, Pd = ctx.saved_tensors
        nh, hd, scale, p_drop, seed = ctx.meta
        dout = _c(dout)
        B, T, d = x.shape
        rows = B * T
        dev = x.device
        datt = H.linear_bwd_data(dout, out_w)
        g_ow = _emit(out_w, lambda o: H.linear_bwd_weight_raw(dout.data_ptr(), d, att.data_ptr(), d, o, rows, d, d),
                     (dout, att))
        g_ob = _emit(out_b, lambda o: H.colsum_raw(dout.data_ptr(), d, o, rows, d), heavy=False)
        if ctx.fused:           # (P is the saved lse2 here)
            dqkv = H.attn_bwd(qkv, att, datt, key_pad, P, nh, scale, p_drop, seed)
            g0 = dqkv.data_ptr()
            dx = H.linear_bwd_data(dqkv, in_w) if ctx.needs_input_grad[0] else None
            g_iw = _emit(in_w, lambda o: H.linear_bwd_weight_raw(g0, 3 * d, x.data_ptr(), d, o, rows, d, 3 * d), (dqkv, x))
            g_ib = _emit(in_b, lambda o: H.colsum_raw(g0, 3 * d, o, rows, 3 * d), heavy=False)
            return dx, None, g_iw, g_ib, g_ow, g_ob, None, None, None
        q0 = qkv.data_ptr()
        dqkv = torch.empty_like(qkv)
        g0 = dqkv.data_ptr()
        Tp = P.shape[-1]
        dP = torch.empty_like(P)
        # dPd = dAtt_h V_h^T
        _bgemm('nt', datt.data_ptr(), d, T * d, hd, q0 + 2 * d * _F4, 3 * d, T * 3 * d, hd, dP.data_ptr(), Tp,
               nh * T * Tp, T * Tp, T, T, hd, B, nh, dev)
        # dV_h = Pd^T dAtt_h
        _bgemm('tn', Pd.data_ptr(), Tp, nh * T * Tp, T * Tp, datt.data_ptr(), d, T * d, hd, g0 + 2 * d * _F4, 3 * d,
               T * 3 * d, hd, T, hd, T, B, nh, dev, padded=True)
        # dPd -> dS: the dropout mask is re-derived inside the softmax gradient kernel
        _lib.call('ft_softmax_bwd', P.data_ptr(), dP.data_ptr(), B, nh, T, T, Tp, scale, p_drop, seed, H._stream())
        # dQ_h = dS K_h ; dK_h = dS^T Q_h
        _bgemm('nn', dP.data_ptr(), Tp, nh * T * Tp, T * Tp, q0 + d * _F4, 3 * d, T * 3 * d, hd, g0, 3 * d, T * 3 * d,
               hd, T, hd, T, B, nh, dev, padded=True)
        _bgemm('tn', dP.data_ptr(), Tp, nh * T * Tp, T * Tp, q0, 3 * d, T * 3 * d, hd, g0 + d * _F4, 3 * d, T * 3 * d,
               hd, T, hd, T, B, nh, dev, padded=True)
        dx = H.linear_bwd_data(dqkv, in_w) if ctx.needs_input_grad[0] else None
        g_iw = _emit(in_w, lambda o: H.linear_bwd_weight_raw(g0, 3 * d, x.data_ptr(), d, o, rows, d, 3 * d), (dqkv, x))
        g_ib = _emit(in_b, lambda o: H.colsum_raw(g0, 3 * d, o, rows, 3 * d), heavy=False)
        return dx, None, g_iw, g_ib, g_ow, g_ob, None, None, None


class AddLayerNormFn(Function):
    """LayerNorm(x + dropout_p(res))  (FFTBlock: src = norm(src + dropout(src2)), common_layers.py:175-176,181-183) in
    one pass: the dropout of the residual branch uses the same counter-based mask as ops.DropoutFn, applied while the
    rows are normalised (no separate dropout pass forward or backward).  res=None: plain LayerNorm
    (ForwardTransformer.norm, :217)."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps, p=0.0, seed=0):
        x = _c(x)
        D = x.shape[-1]
        rows = x.numel() // D
        s = torch.empty_like(x) if res is not None else x
        y = torch.empty_like(x)
        mean = torch.empty(rows, device=x.device, dtype=x.dtype)
        rstd = torch.empty(rows, device=x.device, dtype=x.dtype)
        p = float(p) if res is not None else 0.0
        _lib.call('ft_layernorm_fwd', x.data_ptr(), _p(_c(res) if res is not None else None), gamma.data_ptr(),
                  beta.data_ptr(), s.data_ptr() if res is not None else None, y.data_ptr(), mean.data_ptr(),
                  rstd.data_ptr(), rows, D, float(eps), p, int(seed), H._stream())
        ctx.save_for_backward(s, gamma, beta, mean, rstd)
        ctx.has_res = res is not None
        ctx.p, ctx.seed = p, int(seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        s, gamma, beta, mean, rstd = ctx.saved_tensors
        dy = _c(dy)
        D = s.shape[-1]
        rows = s.numel() // D
        dx = torch.empty_like(s)
        t = torch.empty_like(s)
        dres = torch.empty_like(s) if ctx.has_res and ctx.p > 0 else None
        _lib.call('ft_layernorm_bwd', dy.data_ptr(), s.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                  dx.data_ptr(), t.data_ptr(), _p(dres), rows, D, ctx.p, ctx.seed, H._stream())
        dg, db = _emit_multi([gamma, beta], lambda o: H.colsum2_raw(t.data_ptr(), dy.data_ptr(), D, o[0], o[1], rows, D),
                             heavy=False)
        return dx, ((dres if dres is not None else dx) if ctx.has_res else None), dg, db, None, None, None


class ConvBiasFn(Function):
    """nn.Conv1d(Cin, Cout, k, padding=k//2) with bias (+ ReLU) on channels-last x (FFTBlock conv1/conv2)."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        x = _c(x)
        B, T, Cin = x.shape
        Cout, _, k = w.shape
        wp = H.conv_pack_weight(w)
        y = torch.empty(B, T, Cout, device=x.device, dtype=x.dtype)
        _lib.call('ft_conv1d_bias_fwd', x.data_ptr(), Cin, wp.data_ptr(), b.data_ptr(), y.data_ptr(), Cout, B, T, Cin,
                  Cout, k, int(relu), H._stream())
        ctx.save_for_backward(x, wp, y if relu else None, w, b)
        ctx.relu = bool(relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wp, y, w, b = ctx.saved_tensors
        dy = _c(dy)
        B, T, Cin = x.shape
        Cout = w.shape[0]
        if ctx.relu:
            g = torch.empty_like(dy)
            _lib.call('ft_relu_bwd', dy.data_ptr(), y.data_ptr(), g.data_ptr(), dy.numel(), H._stream())
            dy = g
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            H.conv1d_bwd_data_raw(dy.data_ptr(), Cout, wp, dx, B, T, T, T, False, w=w)
        dw = _emit(w, lambda o: H.conv1d_bwd_weight_raw(dy.data_ptr(), Cout, x, o, T, T), (dy, x))
        db = _emit(b, lambda o: H.colsum_raw(dy.data_ptr(), Cout, o, B * T, Cout), heavy=False)
        return dx, dw, db, None


def check_posenc_length(T: int, pe: torch.Tensor) -> None:
    """the kernels index pe[t] for t < T: a sequence longer than the `pe` buffer (max_len = 5000 positions) must fail
    like the reference's shape error (common_layers.py:144), not read past the buffer"""
    if T > pe.shape[0] or pe.numel() < T * pe.shape[-1]:
        raise _lib.FtError(f'PositionalEncoding: sequence of {T} positions exceeds the pe buffer ({pe.shape[0]})')


class PosEncFn(Function):
    """x + scale * pe[:T]  (PositionalEncoding.forward, common_layers.py:143-145)."""

    @staticmethod
    def forward(ctx, x, pe, scale):
        x = _c(x)
        B, T, D = x.shape
        check_posenc_length(T, pe)
        out = torch.empty_like(x)
        _lib.call('ft_posenc_fwd', x.data_ptr(), pe.data_ptr(), scale.data_ptr(), out.data_ptr(), B, T, D, H._stream())
        ctx.save_for_backward(pe, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        pe, scale = ctx.saved_tensors
        dout = _c(dout)
        B, T, D = dout.shape

        def dscale(o):
            ws = H.workspace(_lib.query('ft_posenc_workspace'), dout.device)
            _lib.call('ft_posenc_bwd_scale', dout.data_ptr(), pe.data_ptr(), o.data_ptr(), B, T, D, ws.data_ptr(),
                      ws.numel(), H._stream())

        return dout, None, _emit(scale, dscale, heavy=False)


# ---------------------------------------------------------------------------------------------------
class PositionalEncoding(nn.Module):
    """common_layers.py:127-145 (buffer `pe` [max_len,1,d] and learnable scalar `scale` kept for the state_dict)."""

    def __init__(self, d_model: int, dropout=0.1, max_len=5000) -> None:
        super().__init__()
        self.p = dropout
        self.scale = nn.Parameter(torch.ones(1))
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer('pe', pe.unsqueeze(0).transpose(0, 1))

    def forward(self, x: torch.Tensor) -> torch.Tensor:            # x [B,T,d]
        if x.shape[1] > self.pe.shape[0]:
            raise _lib.FtError(f'sequence length {x.shape[1]} exceeds PositionalEncoding max_len {self.pe.shape[0]}')
        x = PosEncFn.apply(x, self.pe, self.scale)
        return _dropout(x, self.p, self.training)


class FFTBlock(nn.Module):
    """common_layers.py:148-185 on batch-major [B,T,d]."""

    def __init__(self, d_model: int, nhead: int, conv1_kernel: int, conv2_kernel: int, d_fft: int,
                 dropout: float = 0.1):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.conv1 = nn.Conv1d(in_channels=d_model, out_channels=d_fft, kernel_size=conv1_kernel, stride=1,
                               padding=conv1_kernel // 2)
        self.conv2 = nn.Conv1d(in_channels=d_fft, out_channels=d_model, kernel_size=conv2_kernel, stride=1,
                               padding=conv2_kernel // 2)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.nhead = nhead
        self.p = dropout

    def forward(self, src: torch.Tensor, key_pad: Optional[torch.Tensor] = None) -> torch.Tensor:
        a = self.self_attn
        p = self.p if self.training else 0.0
        seed = _seed() if p > 0 else 0
        src2 = MHAFn.apply(src, key_pad, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias,
                           self.nhead, p, seed)
        # the two F.dropout of the residual branches (common_layers.py:175,182) are fused into the LayerNorm kernels
        src = AddLayerNormFn.apply(src, src2, self.norm1.weight, self.norm1.bias, self.norm1.eps, p,
                                   _seed() if p > 0 else 0)
        src2 = ConvBiasFn.apply(src, self.conv1.weight, self.conv1.bias, True)
        src2 = ConvBiasFn.apply(src2, self.conv2.weight, self.conv2.bias, False)
        return AddLayerNormFn.apply(src, src2, self.norm2.weight, self.norm2.bias, self.norm2.eps, p,
                                    _seed() if p > 0 else 0)


class ForwardTransformer(nn.Module):
    """common_layers.py:188-223; x [B,T,d] -> [B,T,d].  Like the reference, every layer is a deepcopy of ONE
    initialised FFTBlock, so all layers start from identical weights."""

    def __init__(self, d_model: int, d_fft: int, layers: int, heads: int, conv1_kernel: int, conv2_kernel: int,
                 dropout: float = 0.1) -> None:
        super().__init__()
        self.d_model = d_model
        self.pos_encoder = PositionalEncoding(d_model, dropout)
        encoder_layer = FFTBlock(d_model=d_model, nhead=heads, d_fft=d_fft, conv1_kernel=conv1_kernel,
                                 conv2_kernel=conv2_kernel, dropout=dropout)
        encoder_norm = nn.LayerNorm(d_model)
        self.layers = nn.ModuleList([copy.deepcopy(encoder_layer) for _ in range(layers)])
        self.norm = encoder_norm

    def forward(self, x: torch.Tensor, src_pad_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        key_pad = None
        if src_pad_mask is not None:
            key_pad = src_pad_mask.to(torch.uint8).contiguous()
        x = self.pos_encoder(x)
        for layer in self.layers:
            x = layer(x, key_pad)
        return AddLayerNormFn.apply(x, None, self.norm.weight, self.norm.bias, self.norm.eps)


class SeriesPredictor(nn.Module):
    """fast_pitch.py:14-41"""

    def __init__(self, num_chars: int, d_model: int, n_heads: int, d_fft: int, layers: int, conv1_kernel: int,
                 conv2_kernel: int, dropout=0.1):
        super().__init__()
        self.embedding = nn.Embedding(num_chars, d_model)
        self.transformer = ForwardTransformer(heads=n_heads, dropout=dropout, d_model=d_model, d_fft=d_fft,
                                              conv1_kernel=conv1_kernel, conv2_kernel=conv2_kernel, layers=layers)
        self.lin = nn.Linear(d_model, 1)

    def forward(self, x: torch.Tensor, src_pad_mask: Optional[torch.Tensor] = None, alpha: float = 1.0):
        x = ops.EmbeddingFn.apply(x, self.embedding.weight)
        x = self.transformer(x, src_pad_mask=src_pad_mask)
        x = ops.LinearFn.apply(x, self.lin.weight, self.lin.bias)
        if alpha != 1.0:
            x = ops.ScaleFn.apply(x, 1.0 / alpha)
        return x


class FastPitch(nn.Module):
    """Drop-in for models/fast_pitch.py:44-235."""

    def __init__(self, num_chars: int,
                 durpred_dropout: float, durpred_d_model: int, durpred_n_heads: int, durpred_layers: int,
                 durpred_d_fft: int,
                 pitch_dropout: float, pitch_d_model: int, pitch_n_heads: int, pitch_layers: int, pitch_d_fft: int,
                 energy_dropout: float, energy_d_model: int, energy_n_heads: int, energy_layers: int,
                 energy_d_fft: int,
                 pitch_strength: float, energy_strength: float, d_model: int, conv1_kernel: int, conv2_kernel: int,
                 prenet_layers: int, prenet_heads: int, prenet_fft: int, prenet_dropout: float,
                 postnet_layers: int, postnet_heads: int, postnet_fft: int, postnet_dropout: float,
                 n_mels: int, padding_value=PAD_VALUE):
        super().__init__()
        self.padding_value = padding_value
        # 'fp32' (the reference's arithmetic; parity bars) or 'bf16' (BASELINE configs[2]): matmul operands rounded to
        # bf16, fp32 accumulation; LayerNorm / softmax statistics / losses / optimizer stay fp32.  forward() / generate()
        # run under it; trainer.TrainStep extends it over backward (hip.gemm_precision for a hand-rolled backward).
        self.matmul_dtype = 'fp32'
        self.lr = LengthRegulator()
        # predictor branches share no graph node with the trunk in training (trainer.TrainStep may run their backward as a
        # stage of its own)
        self.independent_predictors = True
        self.dur_pred = SeriesPredictor(num_chars=num_chars, d_model=durpred_d_model, n_heads=durpred_n_heads,
                                        layers=durpred_layers, d_fft=durpred_d_fft, conv1_kernel=conv1_kernel,
                                        conv2_kernel=conv2_kernel, dropout=durpred_dropout)
        self.pitch_pred = SeriesPredictor(num_chars=num_chars, d_model=pitch_d_model, n_heads=pitch_n_heads,
                                          layers=pitch_layers, d_fft=pitch_d_fft, conv1_kernel=conv1_kernel,
                                          conv2_kernel=conv2_kernel, dropout=pitch_dropout)
        self.energy_pred = SeriesPredictor(num_chars=num_chars, d_model=energy_d_model, n_heads=energy_n_heads,
                                           layers=energy_layers, d_fft=energy_d_fft, conv1_kernel=conv1_kernel,
                                           conv2_kernel=conv2_kernel, dropout=energy_dropout)
        self.embedding = nn.Embedding(num_embeddings=num_chars, embedding_dim=d_model)
        self.prenet = ForwardTransformer(heads=prenet_heads, dropout=prenet_dropout, conv1_kernel=conv1_kernel,
                                         conv2_kernel=conv2_kernel, d_model=d_model, d_fft=prenet_fft,
                                         layers=prenet_layers)
        self.postnet = ForwardTransformer(heads=postnet_heads, dropout=postnet_dropout, conv1_kernel=conv1_kernel,
                                          conv2_kernel=conv2_kernel, d_model=d_model, d_fft=postnet_fft,
                                          layers=postnet_layers)
        self.lin = nn.Linear(d_model, n_mels)
        self.register_buffer('step', torch.zeros(1, dtype=torch.long))
        self.pitch_strength = pitch_strength
        self.energy_strength = energy_strength
        self.pitch_proj = nn.Conv1d(1, d_model, kernel_size=3, padding=1)
        self.energy_proj = nn.Conv1d(1, d_model, kernel_size=3, padding=1)

    def __repr__(self):
        return f'FastPitch, num params: {sum(p.numel() for p in self.parameters())}'

    def _require_device(self, t: torch.Tensor) -> None:
        if not t.is_cuda or not self.embedding.weight.is_cuda:
            raise _lib.FtError('FastPitch runs on an MI355X (HIP) device only; there is no CPU fallback')

    def _mel(self, x_idx, tok_mask, dur, pitch, energy, frame_lens: Optional[torch.Tensor]):
        x = ops.EmbeddingFn.apply(x_idx, self.embedding.weight)
        x = self.prenet(x, src_pad_mask=tok_mask)
        x = ops.CondAddFn.apply(x, pitch, energy, self.pitch_proj.weight, self.pitch_proj.bias,
                                self.energy_proj.weight, self.energy_proj.bias, self.pitch_strength,
                                self.energy_strength, False)
        x = self.lr(x, dur)
        frame_mask = None
        if frame_lens is not None:          # fast_pitch.py:152-154
            T = x.shape[1]
            frame_mask = torch.arange(T, device=x.device).unsqueeze(0) >= frame_lens.unsqueeze(1)
        x = self.postnet(x, src_pad_mask=frame_mask)
        return ops.LinearFn.apply(x, self.lin.weight, self.lin.bias)           # [B,T,n_mels]

    @precision_scoped
    def forward(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        x = batch['x']
        mel = batch['mel']
        dur = batch['dur']
        mel_lens = batch['mel_len']
        self._require_device(x)
        if self.training:
            self.step += 1
        len_mask = x == 0                                                       # make_token_len_mask
        # the three token-side predictors only feed the losses in training (fast_pitch.py:129-131 vs :133-150): side HIP
        # stream, concurrently with the frame-side trunk (autograd replays their backward on the same stream)
        main = torch.cuda.current_stream()
        side = self._side_stream(x.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            dur_hat = self.dur_pred(x, src_pad_mask=len_mask).squeeze(-1)
            pitch_hat = self.pitch_pred(x, src_pad_mask=len_mask).transpose(1, 2)
            energy_hat = self.energy_pred(x, src_pad_mask=len_mask).transpose(1, 2)
            hook = getattr(self, 'predictor_hook', None)   # trainer.TrainStep: the predictors' losses + backward, right here
            if hook is not None:
                hook({'dur': dur_hat, 'pitch': pitch_hat, 'energy': energy_hat})
        mel_cl = self._mel(x, len_mask, dur, batch['pitch'], batch['energy'],
                           mel_lens.to(device=x.device, dtype=torch.long))
        x_mel = ops.TransposePadFn.apply(mel_cl, mel.size(2), self.padding_value)
        main.wait_stream(side)
        for t in (dur_hat, pitch_hat, energy_hat):
            t.record_stream(main)
        return {'mel': x_mel, 'mel_post': x_mel, 'dur': dur_hat, 'pitch': pitch_hat, 'energy': energy_hat}

    @precision_scoped
    def generate(self, x: torch.Tensor, alpha=1.0,
                 pitch_function: Callable[[torch.Tensor], torch.Tensor] = lambda x: x,
                 energy_function: Callable[[torch.Tensor], torch.Tensor] = lambda x: x) -> Dict[str, torch.Tensor]:
        self.eval()
        with torch.no_grad():
            self._require_device(x)
            B = x.shape[0]
            # NB (reference quirk, fast_pitch.py:174-180): the predictors run WITHOUT a padding mask here
            dur_hat = self.dur_pred(x, alpha=alpha).squeeze(2)
            if torch.sum(dur_hat.long()) <= 0:
                torch.fill_(dur_hat, value=2.)
            pitch_hat = pitch_function(self.pitch_pred(x).transpose(1, 2))
            energy_hat = energy_function(self.energy_pred(x).transpose(1, 2))
            dur_in = dur_hat.contiguous()
            mel_cl = self._mel(x, x == 0, dur_in, pitch_hat.reshape(B, -1).contiguous(),
                               energy_hat.reshape(B, -1).contiguous(), None)
            m = H.transpose_pad_fwd(mel_cl, mel_cl.shape[1], 0.0)
            return {'mel': m, 'mel_post': m, 'dur': dur_in, 'pitch': pitch_hat, 'energy': energy_hat}

    def pad(self, x: torch.Tensor, max_len: int) -> torch.Tensor:
        x = x[:, :, :max_len]
        return torch.nn.functional.pad(x, [0, max_len - x.size(2), 0, 0], 'constant', self.padding_value)

    def _side_stream(self, device) -> 'torch.cuda.Stream':
        key = torch.device(device).index or 0
        if not hasattr(self, '_streams'):
            self._streams = {}
        if key not in self._streams:
            from .model import _side_priority
            self._streams[key] = torch.cuda.Stream(device=device, priority=_side_priority())
        return self._streams[key]

    def get_step(self) -> int:
        return self.step.data.item()

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> 'FastPitch':
        model_config = config['fast_pitch']['model']
        model_config['num_chars'] = config.get('num_chars', NUM_CHARS_DEFAULT)
        model_config['n_mels'] = config['dsp']['num_mels']
        return FastPitch(**model_config)

    @classmethod
    def from_checkpoint(cls, path: Union[Path, str]) -> 'FastPitch':
        checkpoint = torch.load(path, map_location=torch.device('cpu'), weights_only=True)
        model = FastPitch.from_config(checkpoint['config'])
        model.load_state_dict(checkpoint['model'])
        return model
