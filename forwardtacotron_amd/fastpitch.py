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


def mha_fwd(x, key_pad, in_w, in_b, out_w, out_b, nheads, p_drop, seed):
    """nn.MultiheadAttention(d, nheads, dropout)(x, x, x, key_padding_mask=key_pad)[0]  (common_layers.py:172-174) on
    batch-major x [B,T,d]; key_pad uint8 [B,T] (1 = padded key) or None.  -> (out, tape for mha_bwd)"""
    B, T, d = x.shape
    nh = int(nheads)
    hd = d // nh
    scale = 1.0 / math.sqrt(hd)
    qkv = H.linear_fwd(x, in_w, in_b)                                   # [B,T,3d]
    tape = dict(x=x, qkv=qkv, key_pad=key_pad, nh=nh, hd=hd, scale=scale, p=float(p_drop), seed=int(seed))
    # bf16 mode: ONE flash-style kernel between the two projections -- no [B,h,T,T] tensor in memory, the backward
    # recomputes the probabilities (csrc/ft_attn.hip); FT_ATTN_FUSED=0: the five-launch form below in bf16 too
    if H.gemm_precision_mode() == 'bf16' and hd in (64, 128) and os.environ.get('FT_ATTN_FUSED', '1') == '1':
        att, lse2 = H.attn_fwd(qkv, key_pad, nh, scale, p_drop, seed)
        tape.update(fused=True, att=att, lse2=lse2)
        return H.linear_fwd(att, out_w, out_b), tape
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
    tape.update(fused=False, att=att, P=P, Pd=Pd)
    return H.linear_fwd(att, out_w, out_b), tape


def mha_bwd(tape, dout, in_w, in_b, out_w, out_b, need_dx=True, dx_into=None):
    """-> (dx or None, g_in_w, g_in_b, g_out_w, g_out_b); dx_into: an existing gradient of x that dx is ADDED onto"""
    x, qkv, att, key_pad = tape['x'], tape['qkv'], tape['att'], tape['key_pad']
    nh, hd, scale, p_drop, seed = tape['nh'], tape['hd'], tape['scale'], tape['p'], tape['seed']
    B, T, d = x.shape
    rows = B * T
    dev = x.device
    datt = H.linear_bwd_data(dout, out_w)
    g_ow = _emit(out_w, lambda o: H.linear_bwd_weight_raw(dout.data_ptr(), d, att.data_ptr(), d, o, rows, d, d),
                 (dout, att))
    g_ob = _emit(out_b, lambda o: H.colsum_raw(dout.data_ptr(), d, o, rows, d), (dout,), heavy='light')
    if tape['fused']:
        dqkv = H.attn_bwd(qkv, att, datt, key_pad, tape['lse2'], nh, scale, p_drop, seed)
        g0 = dqkv.data_ptr()
    else:
        P, Pd = tape['P'], tape['Pd']
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
    dx = None
    if need_dx:
        dx = H.linear_bwd_data(dqkv, in_w, dx=dx_into, accumulate=dx_into is not None)
    g_iw = _emit(in_w, lambda o: H.linear_bwd_weight_raw(g0, 3 * d, x.data_ptr(), d, o, rows, d, 3 * d), (dqkv, x))
    g_ib = _emit(in_b, lambda o: H.colsum_raw(g0, 3 * d, o, rows, 3 * d), (dqkv,), heavy='light')
    return dx, g_iw, g_ib, g_ow, g_ob


class MHAFn(Function):
    """mha_fwd / mha_bwd as one autograd node (the layer-level entry; models run whole transformers through
    TransformerFn)."""

    @staticmethod
    def forward(ctx, x, key_pad, in_w, in_b, out_w, out_b, nheads, p_drop, seed):
        out, ctx.tape = mha_fwd(_c(x), key_pad, in_w, in_b, out_w, out_b, nheads, p_drop, seed)
        ctx.params = (in_w, in_b, out_w, out_b)
        return out

    @staticmethod
    def backward(ctx, dout):
        dx, g_iw, g_ib, g_ow, g_ob = mha_bwd(ctx.tape, _c(dout), *ctx.params, need_dx=ctx.needs_input_grad[0])
        ctx.tape = None
        return dx, None, g_iw, g_ib, g_ow, g_ob, None, None, None


def addln_fwd(x, res, gamma, beta, eps, p=0.0, seed=0):
    """LayerNorm(x + dropout_p(res)) in one pass -> (y, tape); res None: plain LayerNorm"""
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
    return y, dict(s=s, mean=mean, rstd=rstd, has_res=res is not None, p=p, seed=int(seed))


def addln_bwd(tape, dy, gamma, beta, own_dres=False):
    """-> (dx, dres or None, dgamma, dbeta).  Without dropout the residual branch's gradient IS dx: the same tensor is
    returned twice unless own_dres asks for a copy of its own (a caller that goes on to accumulate into one of them
    while a side-stream weight gradient still reads the other must)."""
    s, mean, rstd = tape['s'], tape['mean'], tape['rstd']
    D = s.shape[-1]
    rows = s.numel() // D
    dx = torch.empty_like(s)
    t = torch.empty_like(s)
    dres = torch.empty_like(s) if tape['has_res'] and (tape['p'] > 0 or own_dres) else None
    _lib.call('ft_layernorm_bwd', dy.data_ptr(), s.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
              dx.data_ptr(), t.data_ptr(), _p(dres), rows, D, tape['p'], tape['seed'], H._stream())
    dg, db = _emit_multi([gamma, beta], lambda o: H.colsum2_raw(t.data_ptr(), dy.data_ptr(), D, o[0], o[1], rows, D),
                         (t, dy), heavy='light')
    return dx, ((dres if dres is not None else dx) if tape['has_res'] else None), dg, db


def convbias_fwd(x, w, b, relu):
    """nn.Conv1d(Cin, Cout, k, padding=k//2) with bias (+ ReLU) on channels-last x -> (y, tape)"""
    B, T, Cin = x.shape
    Cout, _, k = w.shape
    wp = H.conv_pack_weight(w)
    y = torch.empty(B, T, Cout, device=x.device, dtype=x.dtype)
    _lib.call('ft_conv1d_bias_fwd', x.data_ptr(), Cin, wp.data_ptr(), b.data_ptr(), y.data_ptr(), Cout, B, T, Cin,
              Cout, k, int(relu), H._stream())
    return y, dict(x=x, wp=wp, y=y if relu else None, relu=bool(relu))


def convbias_bwd(tape, dy, w, b, need_dx=True, dx_into=None):
    """-> (dx or None, dw, db); dx_into: an existing gradient of x that dx is ADDED onto"""
    x, wp, y = tape['x'], tape['wp'], tape['y']
    B, T, Cin = x.shape
    Cout = w.shape[0]
    if tape['relu']:
        g = torch.empty_like(dy)
        _lib.call('ft_relu_bwd', dy.data_ptr(), y.data_ptr(), g.data_ptr(), dy.numel(), H._stream())
        dy = g
    dx = None
    if need_dx:
        dx = dx_into if dx_into is not None else torch.empty_like(x)
        H.conv1d_bwd_data_raw(dy.data_ptr(), Cout, wp, dx, B, T, T, T, dx_into is not None, w=w)
    dw = _emit(w, lambda o: H.conv1d_bwd_weight_raw(dy.data_ptr(), Cout, x, o, T, T), (dy, x))
    db = _emit(b, lambda o: H.colsum_raw(dy.data_ptr(), Cout, o, B * T, Cout), (dy,), heavy='light')
    return dx, dw, db


class AddLayerNormFn(Function):
    """LayerNorm(x + dropout_p(res))  (FFTBlock: src = norm(src + dropout(src2)), common_layers.py:175-176,181-183) in
    one pass: the dropout of the residual branch uses the same counter-based mask as ops.DropoutFn, applied while the
    rows are normalised (no separate dropout pass forward or backward).  res=None: plain LayerNorm
    (ForwardTransformer.norm, :217)."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps, p=0.0, seed=0):
        y, ctx.tape = addln_fwd(_c(x), res, gamma, beta, eps, p, seed)
        ctx.params = (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        dx, dres, dg, db = addln_bwd(ctx.tape, _c(dy), *ctx.params)
        ctx.tape = None
        return dx, dres, dg, db, None, None, None


class ConvBiasFn(Function):
    """nn.Conv1d(Cin, Cout, k, padding=k//2) with bias (+ ReLU) on channels-last x (FFTBlock conv1/conv2)."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        y, ctx.tape = convbias_fwd(_c(x), w, b, relu)
        ctx.params = (w, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        dx, dw, db = convbias_bwd(ctx.tape, _c(dy), *ctx.params, need_dx=ctx.needs_input_grad[0])
        ctx.tape = None
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
# whole FFTBlocks per C call (include/fwdtaco_hip.h: FtFFTBlock, ft_fft_blocks_fwd / ft_fft_blocks_bwd)
# ---------------------------------------------------------------------------------------------------
import ctypes as _ct

_FP = _ct.c_void_p


class _CBlock(_ct.Structure):
    _fields_ = ([(n, _ct.c_int) for n in ('B', 'T', 'd', 'nheads', 'dfft', 'k1', 'k2')]
                + [(n, _ct.c_float) for n in ('p_drop', 'eps1', 'eps2')]
                + [(n, _ct.c_uint64) for n in ('seed_attn', 'seed_ln1', 'seed_ln2')]
                + [(n, _FP) for n in ('key_pad', 'in_w', 'in_b', 'out_w', 'out_b', 'c1_wp', 'c1_b', 'c2_wp', 'c2_b', 'n1_g',
                                      'n1_b', 'n2_g', 'n2_b', 'in_wT', 'out_wT', 'c1_wpt', 'c2_wpt', 'x', 'qkv', 'att',
                                      'lse2', 'sa', 's1', 'mean1', 'rstd1', 'y1', 'h1', 'h2', 's2', 'mean2', 'rstd2', 'y2')])


class _CGrads(_ct.Structure):
    _fields_ = [(n, _FP) for n in ('dy2', 'dx', 't2', 'd_y1', 'd_h2', 'd_h1', 'g_h1', 't1', 'd_h', 'd_sa', 'datt', 'dqkv',
                                   'g_in_w', 'g_in_b', 'g_out_w', 'g_out_b', 'g_c1_w', 'g_c1_b', 'g_c2_w', 'g_c2_b',
                                   'g_n1_g', 'g_n1_b', 'g_n2_g', 'g_n2_b')]


_sums_streams = {}


def _sums_stream(device) -> 'torch.cuda.Stream':
    key = torch.device(device).index or 0
    if key not in _sums_streams:
        # high priority: twelve small kernels per block -- at default priority they only find free CUs once the other
        # streams' one-wave GEMM grids have drained (the step's join then waited 5 ms for them)
        _sums_streams[key] = torch.cuda.Stream(device=device, priority=-1)
    return _sums_streams[key]


_GRAD_FIELDS = ('g_in_w', 'g_in_b', 'g_out_w', 'g_out_b', 'g_c1_w', 'g_c1_b', 'g_c2_w', 'g_c2_b', 'g_n1_g', 'g_n1_b', 'g_n2_g',
                'g_n2_b')
_side_ws = {}


def _carve(buf: torch.Tensor, sizes):
    """16-byte aligned float sub-buffers of one allocation -> (device addresses, element offsets)"""
    base, off, out, offs = buf.data_ptr(), 0, [], []
    for n in sizes:
        out.append(base + 4 * off)
        offs.append(off)
        off += (n + 3) // 4 * 4
    return out, offs


def _arena(sizes, device):
    return torch.empty(sum((n + 3) // 4 * 4 for n in sizes), device=device, dtype=torch.float32)


def composite_ok(d: int, nheads: int) -> bool:
    """whole blocks per C call: the bf16 mode with the fused attention (head width 64 / 128)"""
    return (H.gemm_precision_mode() == 'bf16' and d % nheads == 0 and d // nheads in (64, 128)
            and os.environ.get('FT_ATTN_FUSED', '1') == '1' and os.environ.get('FT_FFT_COMPOSITE', '1') == '1')


def blocks_fwd_composite(h, key_pad, params, nhead, p, eps1, eps2, seeds):
    """every launch of the transformer's FFTBlocks from ONE C call -> (output of the last block, state for the backward)"""
    B, T, d = h.shape
    R = B * T
    n = len(params) // 12
    f = params[4].shape[0]
    k1, k2 = params[4].shape[2], params[6].shape[2]
    sizes = [3 * R * d, R * d, B * nhead * T, R * d, R * d, R, R, R * d, R * f, R * d, R * d, R, R, R * d]
    names = ('qkv', 'att', 'lse2', 'sa', 's1', 'mean1', 'rstd1', 'y1', 'h1', 'h2', 's2', 'mean2', 'rstd2', 'y2')
    blocks = (_CBlock * n)()
    arenas, keep = [], []
    x_ptr = h.data_ptr()
    for i in range(n):
        in_w, in_b, out_w, out_b, c1w, c1b, c2w, c2b, n1g, n1b, n2g, n2b = params[12 * i:12 * i + 12]
        ar = _arena(sizes, h.device)
        arenas.append(ar)
        addr, offs = _carve(ar, sizes)
        ptrs = dict(zip(names, addr))
        y2_off = offs[-1]
        c1p, c2p = H.conv_pack_weight(c1w), H.conv_pack_weight(c2w)
        keep += [c1p, c2p]
        b = blocks[i]
        b.B, b.T, b.d, b.nheads, b.dfft, b.k1, b.k2 = B, T, d, nhead, f, k1, k2
        b.p_drop, b.eps1, b.eps2 = p, eps1, eps2
        b.seed_attn, b.seed_ln1, b.seed_ln2 = seeds[3 * i], seeds[3 * i + 1], seeds[3 * i + 2]
        b.key_pad = _p(key_pad)
        b.in_w, b.in_b, b.out_w, b.out_b = in_w.data_ptr(), in_b.data_ptr(), out_w.data_ptr(), out_b.data_ptr()
        b.c1_wp, b.c1_b, b.c2_wp, b.c2_b = c1p.data_ptr(), c1b.data_ptr(), c2p.data_ptr(), c2b.data_ptr()
        b.n1_g, b.n1_b, b.n2_g, b.n2_b = n1g.data_ptr(), n1b.data_ptr(), n2g.data_ptr(), n2b.data_ptr()
        b.x = x_ptr
        for k_, v_ in ptrs.items():
            setattr(b, k_, v_)
        x_ptr = ptrs['y2']
    _lib.call('ft_fft_blocks_fwd', _ct.byref(blocks), n, H._stream())
    y = arenas[-1][y2_off:y2_off + R * d].view(B, T, d)          # the last block's output, in place
    return y, dict(blocks=blocks, arenas=arenas, keep=keep, h=h, key_pad=key_pad, dims=(B, T, d, f, k1, k2, nhead))


def blocks_bwd_composite(state, dy, params):
    """-> (d wrt the first block's input, list of the 12 * n parameter gradients (None where a gradient sink took them))"""
    blocks = state['blocks']
    B, T, d, f, k1, k2, nhead = state['dims']
    R = B * T
    n = len(params) // 12
    dev = dy.device
    sizes = [R * d, R * d, R * d, R * f, R * f, R * d, R * d, R * d, R * d, 3 * R * d]
    names = ('t2', 'd_y1', 'd_h2', 'd_h1', 'g_h1', 't1', 'd_h', 'd_sa', 'datt', 'dqkv')
    grads = (_CGrads * n)()
    scratch, outs, sunk = [], [None] * (12 * n), []
    sink = ops._SINK
    dy_ptr = dy.data_ptr()
    last_dh = None
    for i in range(n - 1, -1, -1):
        ps = params[12 * i:12 * i + 12]
        b, g = blocks[i], grads[i]
        wit, wot = H.transpose2d(ps[0]), H.transpose2d(ps[2])
        c1t, c2t = H.conv_pack_weight_t(ps[4]), H.conv_pack_weight_t(ps[6])
        scratch += [wit, wot, c1t, c2t]         # (referenced until the launches that read them have been issued AND run)
        b.in_wT, b.out_wT = wit.data_ptr(), wot.data_ptr()
        b.c1_wpt, b.c2_wpt = c1t.data_ptr(), c2t.data_ptr()
        ar = _arena(sizes, dev)
        scratch.append(ar)
        addr, offs = _carve(ar, sizes)
        ptrs = dict(zip(names, addr))
        for k_, v_ in ptrs.items():
            setattr(g, k_, v_)
        g.dy2, g.dx = dy_ptr, ptrs['d_h']
        dy_ptr = ptrs['d_h']
        last_dh = (ar, offs[names.index('d_h')])
        for j, name in enumerate(_GRAD_FIELDS):
            ent = ops._sink_view(ps[j])
            if ent is not None:
                sink.written.add(ent[0])        # claimed (a second emitter of the same parameter would get a fresh tensor)
                sunk.append(ent[0])
                setattr(g, name, ent[1].data_ptr())
            else:
                t = torch.empty_like(ps[j])
                outs[12 * i + j] = t
                setattr(g, name, t.data_ptr())
    ws = H.workspace(_lib.query('ft_attn_workspace', B, T, nhead), dev)
    side = sink.stream if (sink is not None and sink.stream is not None) else None
    side_raw = side.cuda_stream if side is not None else H._stream()
    def side_buffer(raw, nbytes):
        key = (dev.index or 0, raw)
        buf = _side_ws.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = _side_ws[key] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
        return buf

    wws = side_buffer(side_raw, _lib.query('ft_fft_block_wgrad_workspace', B, T, d, f, k1, k2))
    # FT_FFT_SUMS_STREAM=1: the blocks' bias / LayerNorm column sums on a stream of their own instead of behind the weight-
    # gradient GEMMs, which finish 1.9 ms after the main stream (lab/steady_segments_fp.py).  Measured: 15.9 -> 17.3 ms per
    # step, at default and at high priority alike -- 240 more small launches on a fourth stream cost more than the tail
    # they remove (profiles/r03_fastpitch_ab.txt).  Off.
    sums = _sums_stream(dev) if side is not None and os.environ.get('FT_FFT_SUMS_STREAM', '0') == '1' else None
    sums_raw = sums.cuda_stream if sums is not None else None
    sws = side_buffer(('sums', sums_raw), _lib.query('ft_fft_block_sums_workspace', B, T, d, f)) if sums is not None else None
    _lib.call('ft_fft_blocks_bwd', _ct.byref(blocks), _ct.byref(grads), n, ws.data_ptr(), ws.numel(), wws.data_ptr(),
              wws.numel(), _p(sws), sws.numel() if sws is not None else 0, H._stream(), side_raw, sums_raw)
    if sums is not None:
        sink.used.add(sums)
        if sink.reducer_streams is not None:
            sink.reducer_streams.add(sums)
    dx = last_dh[0][last_dh[1]:last_dh[1] + R * d].view(B, T, d)     # d wrt the first block's input, in place
    if sink is None:
        dx._ft_keep = (scratch, state['arenas'], state['keep'])       # no sink: everything ran on this stream, in order
    if sink is not None:
        sink.keep.append((scratch, state['arenas'], state['keep'], dy))     # read by the weight-gradient stream until joined
        if side is not None:
            sink.used.add(side)
        for idx in sunk:
            sink.used.add(torch.cuda.current_stream())
            if sink.on_write is not None:
                sink.on_write(idx)
    return dx, outs


class TransformerFn(Function):
    """A whole ForwardTransformer (common_layers.py:188-223: positional encoding + dropout, `layers` FFTBlocks, final
    LayerNorm) as ONE autograd node.  Same kernels, same order, same dropout seeds as the per-layer Functions above; what
    goes away is the host side: ~6 autograd nodes per block forward and as many Python backward calls through the
    engine (a FastPitch step was 130 nodes and 14.7 ms of host time for 15 ms of GPU work) -- and the gradient joins of
    the two residual connections, which autograd did with two element-wise adds per block, are now the accumulate
    epilogues of the data-gradient GEMMs that produce the second addend.
    args: x, key_pad, pe, pe_scale, norm_g, norm_b, then 12 tensors per block (in_w, in_b, out_w, out_b, conv1 w, b,
    conv2 w, b, norm1 g, b, norm2 g, b); cfg = (nhead, p_block, p_posenc, training, eps1, eps2, eps_final)."""

    @staticmethod
    def forward(ctx, x, key_pad, pe, pe_scale, norm_g, norm_b, cfg, *params):
        nhead, p_blk, p_pe, training, eps1, eps2, eps_f = cfg
        x = _c(x)
        B, T, D = x.shape
        check_posenc_length(T, pe)
        h = torch.empty_like(x)
        _lib.call('ft_posenc_fwd', x.data_ptr(), pe.data_ptr(), pe_scale.data_ptr(), h.data_ptr(), B, T, D, H._stream())
        pe_seed = 0
        if training and p_pe > 0:
            pe_seed = _seed()
            h = H.dropout(h, p_pe, pe_seed)
        p = p_blk if training else 0.0
        tapes = []
        ctx.comp = None
        if composite_ok(D, nhead) and len(params) >= 12:
            seeds = [(_seed() if p > 0 else 0) for _ in range(3 * (len(params) // 12))]
            h, ctx.comp = blocks_fwd_composite(h, key_pad, params, nhead, p, eps1, eps2, seeds)
            params_loop = ()
        else:
            params_loop = params
        for i in range(len(params_loop) // 12):
            in_w, in_b, out_w, out_b, c1w, c1b, c2w, c2b, n1g, n1b, n2g, n2b = params[12 * i:12 * i + 12]
            s0 = _seed() if p > 0 else 0
            sa, t_mha = mha_fwd(h, key_pad, in_w, in_b, out_w, out_b, nhead, p, s0)
            y1, t_n1 = addln_fwd(h, sa, n1g, n1b, eps1, p, _seed() if p > 0 else 0)
            h1, t_c1 = convbias_fwd(y1, c1w, c1b, True)
            h2, t_c2 = convbias_fwd(h1, c2w, c2b, False)
            h, t_n2 = addln_fwd(y1, h2, n2g, n2b, eps2, p, _seed() if p > 0 else 0)
            tapes.append((t_mha, t_n1, t_c1, t_c2, t_n2))
        y, t_f = addln_fwd(h, None, norm_g, norm_b, eps_f)
        ctx.tapes, ctx.t_f, ctx.params = tapes, t_f, params
        ctx.head = (pe, pe_scale, norm_g, norm_b, p_pe if training else 0.0, pe_seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        pe, pe_scale, norm_g, norm_b, p_pe, pe_seed = ctx.head
        params = ctx.params
        grads = [None] * len(params)
        d, _, g_ng, g_nb = addln_bwd(ctx.t_f, _c(dy), norm_g, norm_b)
        if ctx.comp is not None:
            d, grads = blocks_bwd_composite(ctx.comp, d, params)
            ctx.comp = None
        for i in range(len(ctx.tapes) - 1, -1, -1):
            in_w, in_b, out_w, out_b, c1w, c1b, c2w, c2b, n1g, n1b, n2g, n2b = params[12 * i:12 * i + 12]
            t_mha, t_n1, t_c1, t_c2, t_n2 = ctx.tapes[i]
            with ops.batched_side_launches():       # the block's six weight / six bias gradients: one side-stream hand-over
                d_y1, d_h2, g_n2g, g_n2b = addln_bwd(t_n2, d, n2g, n2b, own_dres=True)
                d_h1, g_c2w, g_c2b = convbias_bwd(t_c2, d_h2, c2w, c2b)
                d_y1, g_c1w, g_c1b = convbias_bwd(t_c1, d_h1, c1w, c1b, dx_into=d_y1)          # + the residual path
                d_h, d_sa, g_n1g, g_n1b = addln_bwd(t_n1, d_y1, n1g, n1b, own_dres=True)
                d, g_iw, g_ib, g_ow, g_ob = mha_bwd(t_mha, d_sa, in_w, in_b, out_w, out_b, dx_into=d_h)   # + the residual path
            grads[12 * i:12 * i + 12] = [g_iw, g_ib, g_ow, g_ob, g_c1w, g_c1b, g_c2w, g_c2b, g_n1g, g_n1b, g_n2g, g_n2b]
        ctx.tapes = None
        if p_pe > 0:
            d = H.dropout(d, p_pe, pe_seed)
        B, T, D = d.shape

        def dscale(o):
            ws = H.workspace(_lib.query('ft_posenc_workspace'), d.device)
            _lib.call('ft_posenc_bwd_scale', d.data_ptr(), pe.data_ptr(), o.data_ptr(), B, T, D, ws.data_ptr(),
                      ws.numel(), H._stream())

        g_scale = _emit(pe_scale, dscale, heavy=False)
        return (d if ctx.needs_input_grad[0] else None, None, None, g_scale, g_ng, g_nb, None, *grads)


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
        # One autograd node per transformer (its FFTBlocks issued from C) in the bf16 mode, where the step was bound by the
        # host; one node per operation in the fp32 mode, which is GPU-bound either way and overlaps better when every
        # weight gradient is side-launched the moment its node runs (25.0 vs 25.5 ms).  FT_TRANSFORMER_NODE=0 / 1 overrides.
        node = os.environ.get('FT_TRANSFORMER_NODE')
        if (node != '1') if node is not None else (H.gemm_precision_mode() != 'bf16'):
            x = self.pos_encoder(x)
            for layer in self.layers:
                x = layer(x, key_pad)
            return AddLayerNormFn.apply(x, None, self.norm.weight, self.norm.bias, self.norm.eps)
        if x.shape[1] > self.pos_encoder.pe.shape[0]:
            raise _lib.FtError(f'sequence length {x.shape[1]} exceeds PositionalEncoding max_len '
                               f'{self.pos_encoder.pe.shape[0]}')
        flat = []
        for l in self.layers:
            a = l.self_attn
            flat += [a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias, l.conv1.weight, l.conv1.bias,
                     l.conv2.weight, l.conv2.bias, l.norm1.weight, l.norm1.bias, l.norm2.weight, l.norm2.bias]
        l0 = self.layers[0]
        cfg = (l0.nhead, l0.p, self.pos_encoder.p, self.training, l0.norm1.eps, l0.norm2.eps, self.norm.eps)
        return TransformerFn.apply(x, key_pad, self.pos_encoder.pe, self.pos_encoder.scale, self.norm.weight,
                                   self.norm.bias, cfg, *flat)


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
