"""Differentiable ops of the ForwardTacotron hot path.

Each op is a torch.autograd.Function whose forward AND backward are sequences of hand-written gfx950
kernels reached through the C ABI (forwardtacotron_amd.hip).  Activations are channels-last [B,T,C]
throughout.  torch only owns the memory, the stream and the autograd graph.
"""
from typing import List, Optional

import torch
from torch.autograd import Function

from . import hip as H

_F4 = 4  # sizeof(float)


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------------------------------------------
class EmbeddingFn(Function):
    """nn.Embedding (forward_tacotron.py:18,73)."""

    _onehot_cache: dict = {}

    @staticmethod
    def forward(ctx, idx, w):
        idx = _c(idx)
        ctx.save_for_backward(idx)
        ctx.V = w.shape[0]
        EmbeddingFn._onehot_cache.clear()      # ids may have changed in place since the last step
        return H.embedding_fwd(idx, w)

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        return None, H.embedding_bwd(idx, _c(dout), ctx.V, EmbeddingFn._onehot_cache)


class LinearFn(Function):
    """nn.Linear over the last dim (forward_tacotron.py:25,100,108 ; common_layers.py:83).
    x_tm_B > 0: x is a TIME-major [T,B,in] recurrence output; the result is batch-major [B,T,out] and the
    gradient handed back to the recurrence is time-major again (no transposition pass either way)."""

    @staticmethod
    def forward(ctx, x, w, b, x_tm_B=0):
        x = _c(x)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        ctx.tmB = int(x_tm_B)
        return H.linear_fwd(x, w, b, x_tm_B=ctx.tmB, y_tm_B=0)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        tmB = ctx.tmB
        dx = H.linear_bwd_data(dy, w, dy_tm_B=0, dx_tm_B=tmB) if ctx.needs_input_grad[0] else None
        if tmB:
            out_f, in_f = w.shape
            rows = x.numel() // in_f
            dw = torch.empty_like(w)
            H.linear_bwd_weight_raw(dy.data_ptr(), out_f, x.data_ptr(), in_f, dw, rows, in_f, out_f, B=tmB,
                                    T=rows // tmB, dy_tm=False, x_tm=True)
        else:
            dw = H.linear_bwd_weight(dy, x)
        db = H.colsum(dy) if ctx.has_bias else None
        return dx, dw, db, None


class BTTransposeFn(Function):
    """[B,T,C] <-> [T,B,C] layout change (only needed when a recurrence output leaves the fused path)."""

    @staticmethod
    def forward(ctx, x, to_time_major):
        ctx.to_tm = bool(to_time_major)
        return H.bt_transpose(_c(x), ctx.to_tm)

    @staticmethod
    def backward(ctx, dy):
        return H.bt_transpose(_c(dy), not ctx.to_tm), None


# ---------------------------------------------------------------------------------------------------
class BatchNormConvFn(Function):
    """BatchNormConv.forward in training mode: conv -> ReLU -> BatchNorm (+ optional residual add,
    common_layers.py:54-57,114).  x [B,T,Cin] -> [B,T,Cout].  Running stats are updated in place."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, residual, running_mean, running_var, relu):
        x = _c(x)
        B, T, Cin = x.shape
        Cout, _, k = w.shape
        Tbuf = T + (1 if k % 2 == 0 else 0)
        wp = H.conv_pack_weight(w)
        y = H.conv1d_fwd(x, wp, relu=relu, Tout=Tbuf)
        out, mean, rstd = H.bn_train_fwd(y, gamma, beta, running_mean, running_var, Tout=T, group=0,
                                         residual=_c(residual) if residual is not None else None)
        ctx.save_for_backward(x, wp, y, gamma, mean, rstd)
        ctx.relu = relu
        ctx.wshape = w.shape
        ctx.has_res = residual is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        x, wp, y, gamma, mean, rstd = ctx.saved_tensors
        dout = _c(dout)
        B, T, Cin = x.shape
        Cout, _, k = ctx.wshape
        Tbuf = y.shape[1]
        dy, dgamma, dbeta = H.bn_bwd(dout, y, gamma, mean, rstd, group=0, relu=ctx.relu)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            H.conv1d_bwd_data_raw(dy.data_ptr(), Cout, wp, dx, B, T, Tbuf, Tbuf, False)
        dw = torch.empty(ctx.wshape, device=x.device, dtype=x.dtype)
        H.conv1d_bwd_weight_raw(dy.data_ptr(), Cout, x, dw, Tbuf, Tbuf)
        dres = dout if ctx.has_res else None
        return dx, dw, dgamma, dbeta, dres, None, None, None


class ConvBankFn(Function):
    """CBHG conv bank + concat + MaxPool1d(2,1,1)[:T]  (common_layers.py:97-105), training mode.
    args: x, K, gamma_cat [K*C], beta_cat, running_mean_cat, running_var_cat (flat storage the per-member
    tensors are views of), then the 3K Parameters w_1..w_K, gamma_1..gamma_K, beta_1..beta_K (autograd
    leaves; gamma_i / beta_i alias slices of the flat storage).   x [B,T,Cin] -> [B,T,K*C]"""

    @staticmethod
    def forward(ctx, x, K, gamma, beta, running_mean, running_var, *params):
        x = _c(x)
        ws = params[:K]
        B, T, Cin = x.shape
        C = ws[0].shape[0]
        wp_all = torch.empty(C * Cin * K * (K + 1) // 2, device=x.device, dtype=x.dtype)
        off = 0
        for i, w in enumerate(ws):
            n = (i + 1) * C * Cin
            H.conv_pack_weight(w, out=wp_all[off:off + n].view(i + 1, C, Cin))
            off += n
        ybank = H.conv_bank_fwd(x, wp_all, K, C, relu=True, Tout=T + 1)
        z, mean, rstd = H.bn_train_fwd(ybank, gamma, beta, running_mean, running_var, Tout=T, group=C)
        out = H.maxpool2_fwd(z)
        ctx.save_for_backward(x, wp_all, ybank, z, gamma, mean, rstd)
        ctx.K, ctx.C = K, C
        return out

    @staticmethod
    def backward(ctx, dout):
        x, wp_all, ybank, z, gamma, mean, rstd = ctx.saved_tensors
        K, C = ctx.K, ctx.C
        B, T, Cin = x.shape
        dz = H.maxpool2_bwd(_c(dout), z)
        dy, dgamma, dbeta = H.bn_bwd(dz, ybank, gamma, mean, rstd, group=C, relu=True)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dws = []
        off = 0
        for i in range(K):
            k = i + 1
            n = k * C * Cin
            Tvalid = T + (1 if k % 2 == 0 else 0)
            dptr = dy.data_ptr() + i * C * _F4
            wp = wp_all[off:off + n].view(k, C, Cin)
            if dx is not None:
                H.conv1d_bwd_data_raw(dptr, K * C, wp, dx, B, T, T + 1, Tvalid, i > 0)
            dw = torch.empty(C, Cin, k, device=x.device, dtype=x.dtype)
            H.conv1d_bwd_weight_raw(dptr, K * C, x, dw, T + 1, Tvalid)
            dws.append(dw)
            off += n
        dgs = [dgamma[i * C:(i + 1) * C] for i in range(K)]
        dbs = [dbeta[i * C:(i + 1) * C] for i in range(K)]
        return (dx, None, None, None, None, None, *dws, *dgs, *dbs)


class DropoutFn(Function):
    """F.dropout(training=True) with a counter-based mask (no mask tensor)."""

    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.p, ctx.seed = float(p), int(seed)
        return H.dropout(_c(x), ctx.p, ctx.seed)

    @staticmethod
    def backward(ctx, dout):
        return H.dropout(_c(dout), ctx.p, ctx.seed), None, None


class ScaleFn(Function):
    @staticmethod
    def forward(ctx, x, s):
        ctx.s = float(s)
        return H.scale(_c(x), ctx.s)

    @staticmethod
    def backward(ctx, dout):
        return H.scale(_c(dout), ctx.s), None


class HighwayFn(Function):
    """HighwayNetwork.forward (common_layers.py:35-40)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        x = _c(x)
        x12 = H.linear_multi_fwd(x, [w1, w2], [b1, b2])
        ctx.save_for_backward(x, x12, w1, w2)
        return H.highway_gate_fwd(x12, x)

    @staticmethod
    def backward(ctx, dout):
        x, x12, w1, w2 = ctx.saved_tensors
        C = x.shape[-1]
        rows = x.numel() // C
        d12, dx = H.highway_gate_bwd(_c(dout), x12, x)
        p = d12.data_ptr()
        H.linear_bwd_data_raw(p, 2 * C, w1, dx, rows, C, True)
        H.linear_bwd_data_raw(p + C * _F4, 2 * C, w2, dx, rows, C, True)
        dw1 = torch.empty_like(w1)
        dw2 = torch.empty_like(w2)
        H.linear_bwd_weight_raw(p, 2 * C, x.data_ptr(), C, dw1, rows, C, C)
        H.linear_bwd_weight_raw(p + C * _F4, 2 * C, x.data_ptr(), C, dw2, rows, C, C)
        db = H.colsum(d12)
        return dx, dw1, db[:C], dw2, db[C:]


# ---------------------------------------------------------------------------------------------------
def _rnn_param_grads(dxp, dhp, x, hid, G, Hh, w_ih_f, w_ih_r, need_dx):
    """Shared tail of the GRU/LSTM backward: weight / bias / input gradients from the per-step
    pre-activation gradients (dxp wrt input projection, dhp wrt hidden projection).
    x is batch-major [B,T,I]; dxp / dhp / hid are the recurrence's time-major [T,B,*] buffers."""
    B, T, I = x.shape
    rows = B * T
    GH = G * Hh
    grads = []
    dx = torch.empty_like(x) if need_dx else None
    dbx = H.colsum(dxp)                       # [2*G*H]
    dbh = dbx if dhp is dxp else H.colsum(dhp)
    for d, w_ih in enumerate((w_ih_f, w_ih_r)):
        px = dxp.data_ptr() + d * GH * _F4
        ph = dhp.data_ptr() + d * GH * _F4
        dw_ih = torch.empty(GH, I, device=x.device, dtype=x.dtype)
        H.linear_bwd_weight_raw(px, 2 * GH, x.data_ptr(), I, dw_ih, rows, I, GH, B=B, T=T, dy_tm=True, x_tm=False)
        dw_hh = torch.empty(GH, Hh, device=x.device, dtype=x.dtype)
        H.linear_bwd_weight_raw(ph, 2 * GH, hid.data_ptr() + d * Hh * _F4, 2 * Hh, dw_hh, rows, Hh, GH, B=B, T=T,
                                x_shift=-1 if d == 0 else 1, dy_tm=True, x_tm=True)
        if dx is not None:
            H.linear_bwd_data_raw(px, 2 * GH, w_ih, dx, rows, GH, d > 0, dy_tm_B=B, dx_tm_B=0)
        grads.append((dw_ih, dw_hh, dbx[d * GH:(d + 1) * GH], dbh[d * GH:(d + 1) * GH]))
    return dx, grads


class BiGRUFn(Function):
    """nn.GRU(bidirectional=True, batch_first=True), h0=0 (common_layers.py:89,123).
    x is batch-major [B,T,I]; the result (and the gradient it expects) is TIME-major [T,B,2H]."""

    @staticmethod
    def forward(ctx, x, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r):
        x = _c(x)
        Hh = w_hh_f.shape[1]
        train = any(ctx.needs_input_grad)
        xp = H.linear_multi_fwd(x, [w_ih_f, w_ih_r], [b_ih_f, b_ih_r], y_tm_B=x.shape[0])
        out, gates = H.gru_fwd(xp, w_hh_f, w_hh_r, b_hh_f, b_hh_r, Hh, save_gates=train)
        if train:
            ctx.save_for_backward(x, out, gates, w_ih_f, w_hh_f, w_ih_r, w_hh_r)
        ctx.Hh = Hh
        return out

    @staticmethod
    def backward(ctx, dout):
        x, out, gates, w_ih_f, w_hh_f, w_ih_r, w_hh_r = ctx.saved_tensors
        Hh = ctx.Hh
        dxp, dhp = H.gru_bwd(_c(dout), out, gates, H.transpose2d(w_hh_f), H.transpose2d(w_hh_r), Hh)
        dx, g = _rnn_param_grads(dxp, dhp, x, out, 3, Hh, w_ih_f, w_ih_r, ctx.needs_input_grad[0])
        return (dx, g[0][0], g[0][1], g[0][2], g[0][3], g[1][0], g[1][1], g[1][2], g[1][3])


class BiLSTMFn(Function):
    """pack_padded_sequence -> nn.LSTM(bidirectional) -> pad_packed_sequence(padding_value)
    (forward_tacotron.py:147-152); lens=None runs over the padded length (:224)."""

    @staticmethod
    def forward(ctx, x, lens, pad_value, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r):
        x = _c(x)
        Hh = w_hh_f.shape[1]
        train = any(ctx.needs_input_grad)
        xp = H.linear_multi_fwd(x, [w_ih_f, w_ih_r], [b_ih_f, b_ih_r], y_tm_B=x.shape[0])
        raw, cst, gates = H.lstm_fwd(xp, w_hh_f, w_hh_r, b_hh_f, b_hh_r, lens, Hh, save_gates=train)
        if train:
            ctx.save_for_backward(x, raw, cst, gates, w_ih_f, w_hh_f, w_ih_r, w_hh_r, lens)
        ctx.Hh = Hh
        ctx.has_lens = lens is not None
        # time-major raw states -> batch-major output, padding_value beyond each item's length
        return H.fill_padded(raw, lens, float(pad_value))

    @staticmethod
    def backward(ctx, dout):
        x, raw, cst, gates, w_ih_f, w_hh_f, w_ih_r, w_hh_r, lens = ctx.saved_tensors
        Hh = ctx.Hh
        dg = H.lstm_bwd(H.bt_transpose(_c(dout), True), raw, cst, gates, H.transpose2d(w_hh_f),
                        H.transpose2d(w_hh_r), lens if ctx.has_lens else None, Hh)
        dx, g = _rnn_param_grads(dg, dg, x, raw, 4, Hh, w_ih_f, w_ih_r, ctx.needs_input_grad[0])
        return (dx, None, None, g[0][0], g[0][1], g[0][2], g[0][3], g[1][0], g[1][1], g[1][2], g[1][3])


# ---------------------------------------------------------------------------------------------------
class CondAddFn(Function):
    """x + pitch_proj(pitch)*s_p + energy_proj(energy)*s_e  (forward_tacotron.py:137-143)."""

    @staticmethod
    def forward(ctx, x, pitch, energy, wp, bp, we, be, sp, se, x_time_major=False):
        x, pitch, energy = _c(x), _c(pitch), _c(energy)
        ctx.save_for_backward(pitch, energy)
        ctx.sp, ctx.se = float(sp), float(se)
        ctx.C = x.shape[-1]
        ctx.x_tm = bool(x_time_major)
        return H.cond_add_fwd(x, pitch, energy, wp, bp, we, be, float(sp), float(se), ctx.x_tm)

    @staticmethod
    def backward(ctx, dout):
        pitch, energy = ctx.saved_tensors
        dout = _c(dout)
        C = ctx.C
        taps = H.cond_taps(pitch, energy)                     # [B,T,8]
        g = H.linear_bwd_weight(dout, taps)                   # [C,8]
        dwp = (g[:, 0:3] * ctx.sp).reshape(C, 1, 3)
        dbp = g[:, 3] * ctx.sp
        dwe = (g[:, 4:7] * ctx.se).reshape(C, 1, 3)
        dbe = g[:, 7] * ctx.se
        dx = H.bt_transpose(dout, True) if ctx.x_tm else dout
        return dx, None, None, dwp, dbp, dwe, dbe, None, None, None


class LengthRegulateFn(Function):
    """LengthRegulator.forward (common_layers.py:17-24).  Clamps `dur` in place like the reference.
    One host sync reads max_b sum_j r_bj to size the output (the reference syncs here too)."""

    @staticmethod
    def forward(ctx, x, dur):
        x = _c(x)
        cum, total = H.lr_scan(dur)
        Tm = int(total.max().item()) if total.numel() else 0
        ctx.save_for_backward(cum)
        ctx.Tx = x.shape[1]
        return H.lr_expand(x, cum, Tm)

    @staticmethod
    def backward(ctx, dy):
        (cum,) = ctx.saved_tensors
        return H.lr_bwd(_c(dy), cum, ctx.Tx), None


class TransposePadFn(Function):
    """[B,T,C] -> [B,C,Tout] + ForwardTacotron._pad (forward_tacotron.py:155,161-162,236-239)."""

    @staticmethod
    def forward(ctx, x, Tout, pad):
        x = _c(x)
        ctx.T = x.shape[1]
        return H.transpose_pad_fwd(x, int(Tout), float(pad))

    @staticmethod
    def backward(ctx, dout):
        return H.transpose_pad_bwd(_c(dout), ctx.T), None, None


class MaskedL1Fn(Function):
    """MaskedL1.forward (trainer/common.py:71-78) on [B,C,T]."""

    @staticmethod
    def forward(ctx, x, target, lens):
        x, target, lens = _c(x), _c(target), _c(lens)
        loss, inv = H.masked_l1_fwd(x, target, lens)
        ctx.save_for_backward(x, target, lens, inv)
        return loss

    @staticmethod
    def backward(ctx, g):
        x, target, lens, inv = ctx.saved_tensors
        g = _c(g).reshape(1).float()
        return H.masked_l1_bwd(x, target, lens, inv, g), None, None


def masked_l1(x, target, lens):
    return MaskedL1Fn.apply(x, target, lens)
