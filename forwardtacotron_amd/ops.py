"""Differentiable ops of the ForwardTacotron hot path.

Each op is a torch.autograd.Function whose forward AND backward are sequences of hand-written gfx950
kernels reached through the C ABI (forwardtacotron_amd.hip).  Activations are channels-last [B,T,C]
throughout; recurrence outputs are time-major [T,B,C].  torch only owns the memory, the streams and
the autograd graph.
"""
from typing import List, Optional

import os
import torch
from torch.autograd import Function

from . import hip as H

_F4 = 4  # sizeof(float)


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------------------------------------------
# Gradient sink (installed by trainer.TrainStep; absent = plain autograd semantics).
#
# Every parameter of the model receives exactly one gradient contribution per backward pass, so the
# weight-gradient kernels can write STRAIGHT into the trainer's flat gradient buffer (no temporary, no
# AccumulateGrad add) and the backward returns None for that input.  Weight gradients are off the
# critical data-gradient chain, so with a sink the GEMM-shaped ones are issued on a second HIP stream and
# overlap the latency-bound persistent recurrences; the trainer joins that stream before the optimiser.
# ---------------------------------------------------------------------------------------------------
class GradSink:
    def __init__(self, views, stream=None, on_write=None):
        self.views = views              # {param.data_ptr(): (index, grad view tensor)}
        self.written = set()
        self.stream = stream            # side stream for weight-gradient GEMMs (None: current stream)
        self.on_write = on_write        # callback(index) -> lets the bucketed all-reduce count arrivals
        # weight-gradient GEMMs whose operands have at most this many rows stay on the CURRENT stream (0: none do).
        # ForwardTacotron sets it to its token-side row count: those gradients (prenet, predictors) only become
        # computable at the very end of the backward, right after the first LSTM's four big weight gradients landed
        # on the side stream -- left there they queue behind them while the main stream has nothing left to do
        # (1.9 ms of a 28.4 ms step); on the main stream the two tails overlap.  FastPitch, whose whole token side
        # is launch-bound, is faster with everything on the side stream and leaves it at 0.
        self.inline_rows = 0
        # defer = True: side-stream weight gradients are not launched where they are emitted but queued, and launched
        # in one go right before the next recurrence's BPTT kernel (ops.flush_deferred) -- the persistent recurrences
        # leave most of the chip idle, while between them the main stream's own GEMM / BatchNorm chain wants it all
        self.defer = False
        self.pending = []               # (compute, views, deps, indices, ready event)
        self.held = set()               # indices queued in `pending`: claimed, not yet issued (the all-reduce must wait)
        self.main = None
        self.used = set()               # streams gradient writes were issued on this step
        # operands of side-stream launches, kept REFERENCED until the step has joined the side stream.  record_stream only
        # keeps their memory from being recycled; it does not keep autograd from ADDING INTO them: a gradient tensor handed
        # to two consumers (LayerNorm's dx for branch and residual, AddBackward's grad for both addends) is accumulated
        # in place -- on the emitting stream -- as soon as it is uniquely owned, i.e. right after the conv / linear node
        # whose weight gradient is still reading it on the side stream has returned (seen as rare wrong conv2 weight
        # gradients in MultiFastPitch's predictors).  A live reference makes autograd add out of place instead.
        self.keep = []
        # late_ok (set per step by the trainer): weight gradients emitted with late=True are not sent to the side stream but
        # run on the step's MAIN stream at the very end of the backward (flush_deferred) -- with the predictors' stage out of
        # the tail the main stream finishes its chain 0.8 ms before the side stream has worked off the LSTM's four weight
        # gradients; one of the four rebalances the two.
        self.late_ok = False
        self.late = []
        # set by the trainer: the set of streams a gradient bucket's all-reduce waits for; whoever writes gradients on yet
        # another stream adds it here
        self.reducer_streams = None

    def begin_step(self):
        self.written.clear()
        self.late.clear()
        self.pending.clear()
        self.held.clear()
        self.used.clear()
        self.keep.clear()
        self.main = torch.cuda.current_stream() if torch.cuda.is_available() else None   # the step's critical stream


_SINK: Optional[GradSink] = None


def set_grad_sink(sink: Optional[GradSink]) -> None:
    global _SINK
    _SINK = sink


def _sink_view(w: torch.Tensor):
    """(index, view) of w's slot in the flat gradient buffer if a sink is installed and the slot is still
    unwritten this step, else None."""
    if _SINK is None:
        return None
    ent = _SINK.views.get(w.data_ptr())
    if ent is None or ent[0] in _SINK.written:
        return None
    return ent


def _sink_done(idx: int) -> None:
    _SINK.written.add(idx)
    # The write was just issued on (or handed over from) the current stream.  Autograd only joins the streams on which
    # it accumulated a leaf gradient itself, and the sink's parameters never get there, so whoever applies the
    # gradients must wait for every stream recorded here (a predictor's backward runs on the predictor side stream).
    _SINK.used.add(torch.cuda.current_stream())
    if _SINK.on_write is not None:
        _SINK.on_write(idx)


def _side_launch(compute, views, deps, idxs) -> None:
    """run compute(views) on the sink's side stream now, or queue it (GradSink.defer)"""
    if _SINK.defer:
        for i in idxs:
            _SINK.written.add(i)        # the slot is taken; the reducer hears about it when the launch is issued
            _SINK.held.add(i)
        # The operands are complete once the EMITTING stream gets here, and the flush may run on another stream (a
        # predictor's recurrence on its side stream flushes gradients the main stream emitted): the item remembers its
        # stream and the flush makes the side stream wait for that stream's position AT FLUSH TIME -- one join per
        # (flush, emitting stream), however many items; a variant with one event per item measured 1.2 ms slower.
        _SINK.pending.append((compute, views, deps, idxs, torch.cuda.current_stream()))
        return
    side = _SINK.stream
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        compute(views)
    # the operands stay referenced until the step has joined the side stream (GradSink.keep): nothing can recycle or
    # overwrite them under the side stream, and no record_stream bookkeeping is needed
    _SINK.keep.append(deps)
    for i in idxs:
        _sink_done(i)


def flush_begin():
    """First half of a flush: joins the side stream with the emitting streams NOW (so that it does not wait for whatever
    the caller launches next) and hands back the queued launches; flush_end issues them.  A recurrence's backward calls
    flush_begin, launches its BPTT kernel, then flush_end: the persistent kernel (all of whose workgroups must become
    co-resident) is dispatched before the GEMMs that would otherwise fill every CU's registers first."""
    sink = _SINK
    if sink is None or not sink.pending:
        return None
    side = sink.stream
    pend, sink.pending = sink.pending, []
    seen = []
    for _, _, _, _, st in pend:
        if st not in seen:
            seen.append(st)
            side.wait_stream(st)
    return pend


def flush_end(pend) -> None:
    if not pend:
        return
    sink = _SINK
    side = sink.stream
    with torch.cuda.stream(side):
        for compute, views, deps, idxs, _ in pend:
            compute(views)
            sink.keep.append(deps)
    for _, _, _, idxs, _ in pend:
        for i in idxs:
            sink.held.discard(i)
            if sink.on_write is not None:
                sink.on_write(i)


class batched_side_launches:
    """`with batched_side_launches():` -- the side-stream weight gradients (and 'light' bias-gradient sums) emitted inside
    are queued and issued together at the end: ONE stream join and ONE stream switch for the lot instead of one per
    parameter (a FastPitch step emitted 88 of them one by one: 3.2 ms of host time in torch's stream calls).  No-op
    without a sink or when the sink defers already (recurrent models flush before their BPTT kernels instead)."""

    def __enter__(self):
        self.on = _SINK is not None and not _SINK.defer
        if self.on:
            _SINK.defer = True
        return self

    def __exit__(self, *exc):
        if self.on:
            flush_end(flush_begin())
            _SINK.defer = False
        return False


def flush_deferred() -> None:
    """issue every queued side-stream weight gradient (no-op without a deferring sink), then the ones kept for the end of
    the main stream's chain (GradSink.late)"""
    flush_end(flush_begin())
    sink = _SINK
    if sink is not None and sink.late:
        late, sink.late = sink.late, []
        for compute, view, deps, idx in late:
            compute(view)
            sink.keep.append(deps)
            sink.held.discard(idx)
            sink.used.add(torch.cuda.current_stream())
            if sink.on_write is not None:
                sink.on_write(idx)


def _emit(w: torch.Tensor, compute, deps=(), heavy=True, late: bool = False, inline_ok: bool = True):
    """Produces the gradient of parameter `w`: compute(out) must overwrite `out` (same shape as w).
    Without a sink: returns a fresh tensor (autograd accumulates it).  With a sink: writes the flat-buffer
    view (on the side stream when `heavy`) and returns None.  heavy='light': a small reduction (bias gradient) that nothing
    on the step's critical stream waits for -- it joins the queued side-stream launches of a deferring sink (22 column
    sums + finalizes = 0.4 ms of the ForwardTacotron step's main stream) and stays inline otherwise."""
    ent = _sink_view(w)
    if ent is None:
        out = torch.empty_like(w)
        compute(out)
        return out
    idx, view = ent
    if late and _SINK.late_ok and _SINK.defer:
        _SINK.written.add(idx)
        _SINK.held.add(idx)             # claimed; the all-reduce hears about it when it is issued
        _SINK.late.append((compute, view, deps, idx))
        return None
    if heavy == 'light':
        heavy = _SINK.defer and bool(deps) and os.environ.get('FT_BIAS_GRADS_SIDE', '1') == '1'
    side = _SINK.stream if heavy else None
    if side is not None and inline_ok and deps and deps[0].numel() // deps[0].shape[-1] <= _SINK.inline_rows:
        side = None                     # short (token-side) operands: see GradSink.inline_rows
    if side is None:
        compute(view)
        _sink_done(idx)
    else:
        _side_launch(compute, view, deps, (idx,))
    return None


def _emit_multi(ws, compute, deps=(), heavy=True):
    """_emit for several parameters whose gradients one launch produces together: compute(outs).  heavy='light': as in
    _emit (joins a batch of queued side-stream launches, inline otherwise)."""
    ents = [_sink_view(w) for w in ws]
    if any(e is None for e in ents):
        outs = [torch.empty_like(w) for w in ws]
        compute(outs)
        return outs
    if heavy == 'light':
        heavy = _SINK.defer and bool(deps) and os.environ.get('FT_BIAS_GRADS_SIDE', '1') == '1'
    side = _SINK.stream if heavy else None
    if side is not None and deps and deps[0].numel() // deps[0].shape[-1] <= _SINK.inline_rows:
        side = None
    views = [e[1] for e in ents]
    if side is None:
        compute(views)
        for e in ents:
            _sink_done(e[0])
    else:
        _side_launch(compute, views, deps, tuple(e[0] for e in ents))
    return [None] * len(ws)


def _emit_copies(ws, values):
    """_emit_copy for several parameters at once: one launch for all the slots the sink still has open."""
    ents = [_sink_view(w) for w in ws]
    outs = [v if e is None else None for e, v in zip(ents, values)]
    todo = [(e, v) for e, v in zip(ents, values) if e is not None]
    if todo:
        H.copy_segments([v for _, v in todo], [e[1] for e, _ in todo])
        for e, _ in todo:
            _sink_done(e[0])
    return outs


def _emit_copy(w: torch.Tensor, value: torch.Tensor):
    """Small vector gradients that a fused kernel already produced in `value`."""
    ent = _sink_view(w)
    if ent is None:
        return value
    ent[1].copy_(value.view_as(ent[1]))
    _sink_done(ent[0])
    return None


# ---------------------------------------------------------------------------------------------------
class EmbeddingFn(Function):
    """nn.Embedding (forward_tacotron.py:18,73)."""
    _onehot_cache: dict = {}

    @staticmethod
    def forward(ctx, idx, w):
        idx = _c(idx)
        ctx.save_for_backward(idx, w)
        EmbeddingFn._onehot_cache.clear()      # ids may have changed in place since the last step
        return H.embedding_fwd(idx, w)

    @staticmethod
    def backward(ctx, dout):
        idx, w = ctx.saved_tensors
        dout = _c(dout)
        V = w.shape[0]
        dw = _emit(w, lambda out: H.embedding_bwd(idx, dout, V, EmbeddingFn._onehot_cache, out=out), (dout, idx))
        return None, dw


class LinearFn(Function):
    """nn.Linear over the last dim (forward_tacotron.py:25,100,108 ; common_layers.py:83).
    x_tm_B > 0: x is a TIME-major [T,B,in] recurrence output; the result is batch-major [B,T,out] and the
    gradient handed back to the recurrence is time-major again (no transposition pass either way)."""

    @staticmethod
    def forward(ctx, x, w, b, x_tm_B=0):
        x = _c(x)
        ctx.save_for_backward(x, w, b)
        ctx.tmB = int(x_tm_B)
        return H.linear_fwd(x, w, b, x_tm_B=ctx.tmB, y_tm_B=0)

    @staticmethod
    def backward(ctx, dy):
        x, w, b = ctx.saved_tensors
        dy = _c(dy)
        tmB = ctx.tmB
        dx = H.linear_bwd_data(dy, w, dy_tm_B=0, dx_tm_B=tmB) if ctx.needs_input_grad[0] else None
        out_f, in_f = w.shape
        rows = x.numel() // in_f

        def wgrad(out):
            if tmB:
                H.linear_bwd_weight_raw(dy.data_ptr(), out_f, x.data_ptr(), in_f, out, rows, in_f, out_f, B=tmB,
                                        T=rows // tmB, dy_tm=False, x_tm=True)
            else:
                H.linear_bwd_weight_raw(dy.data_ptr(), out_f, x.data_ptr(), in_f, out, rows, in_f, out_f)

        dw = _emit(w, wgrad, (dy, x))
        db = None
        if b is not None:
            db = _emit(b, lambda out: H.colsum_raw(dy.data_ptr(), out_f, out, rows, out_f), (dy,), heavy='light')
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
        # the statistics come out of the conv's own GEMM epilogue (128x128 launches) -- no read-back of y for them
        y, part, nch = H.conv1d_fwd_stats(x, wp, relu=relu, Tout=Tbuf)
        out, mean, rstd = H.bn_train_from_partials(part, nch, y, gamma, beta, running_mean, running_var, Tout=T,
                                                   group=0, residual=_c(residual) if residual is not None else None)
        ctx.save_for_backward(x, wp, y, gamma, mean, rstd, w, beta)
        ctx.relu = relu
        ctx.has_res = residual is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        x, wp, y, gamma, mean, rstd, w, beta = ctx.saved_tensors
        dout = _c(dout)
        B, T, Cin = x.shape
        Cout, _, k = w.shape
        Tbuf = y.shape[1]
        eg, eb = _sink_view(gamma), _sink_view(beta)
        dy, dgamma, dbeta = H.bn_bwd(dout, y, gamma, mean, rstd, group=0, relu=ctx.relu,
                                     dgamma=eg[1] if eg else None, dbeta=eb[1] if eb else None)
        if eg:
            _sink_done(eg[0])
            dgamma = None
        if eb:
            _sink_done(eb[0])
            dbeta = None
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            H.conv1d_bwd_data_raw(dy.data_ptr(), Cout, wp, dx, B, T, Tbuf, Tbuf, False, w=w)
        dw = _emit(w, lambda out: H.conv1d_bwd_weight_raw(dy.data_ptr(), Cout, x, out, Tbuf, Tbuf), (dy, x))
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
        wp_all = H.bank_packs(ws, False)
        ybank, part, nch = H.conv_bank_fwd_stats(x, wp_all, K, C, relu=True)
        # BatchNorm apply + MaxPool in one pass over the bank buffer: the normalised tensor z is neither written nor kept
        # (postnet: 220 MB), the backward recomputes the pooling decisions from ybank (FT_BN_POOL_FUSED=0: three passes)
        ctx.fused = H.bn_pool_fusable(ybank, C) and os.environ.get('FT_BN_POOL_FUSED', '1') == '1'
        if ctx.fused:
            out, mean, rstd = H.bn_pool_from_partials(part, nch, ybank, gamma, beta, running_mean, running_var, Tout=T,
                                                      group=C)
            z = beta
        else:
            z, mean, rstd = H.bn_train_from_partials(part, nch, ybank, gamma, beta, running_mean, running_var, Tout=T,
                                                     group=C)
            out = H.maxpool2_fwd(z)
        ctx.save_for_backward(x, wp_all, ybank, z, gamma, mean, rstd, *params)
        ctx.K, ctx.C = K, C
        return out

    @staticmethod
    def backward(ctx, dout):
        x, wp_all, ybank, z, gamma, mean, rstd = ctx.saved_tensors[:7]
        params = ctx.saved_tensors[7:]
        K, C = ctx.K, ctx.C
        ws, gs, bs = params[:K], params[K:2 * K], params[2 * K:3 * K]
        B, T, Cin = x.shape
        if ctx.fused:                     # (the `z` slot holds beta)
            dy, dgamma, dbeta = H.bn_pool_bwd(_c(dout), ybank, gamma, z, mean, rstd, group=C, relu=True)
        else:
            dz = H.maxpool2_bwd(_c(dout), z)
            dy, dgamma, dbeta = H.bn_bwd(dz, ybank, gamma, mean, rstd, group=C, relu=True)
        dx = H.conv_bank_bwd_data(dy, wp_all, K, C, Cin, T, ws=ws) if ctx.needs_input_grad[0] else None
        if C % 128 == 0:        # all members' weight gradients in one launch
            dws = _emit_multi(ws, lambda outs: H.conv_bank_bwd_weight(dy, x, outs, C), (dy, x))
        else:
            dws = []
            for i in range(K):
                k = i + 1
                Tvalid = T + (1 if k % 2 == 0 else 0)
                dptr = dy.data_ptr() + i * C * _F4
                dws.append(_emit(ws[i], lambda out, dptr=dptr, Tvalid=Tvalid: H.conv1d_bwd_weight_raw(
                    dptr, K * C, x, out, T + 1, Tvalid), (dy, x)))
        both = _emit_copies(list(gs) + list(bs), [dgamma[i * C:(i + 1) * C] for i in range(K)] +
                            [dbeta[i * C:(i + 1) * C] for i in range(K)])
        dgs, dbs = both[:K], both[K:]
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
        ctx.save_for_backward(x, x12, w1, w2, b1, b2)
        return H.highway_gate_fwd(x12, x)

    @staticmethod
    def backward(ctx, dout):
        x, x12, w1, w2, b1, b2 = ctx.saved_tensors
        C = x.shape[-1]
        rows = x.numel() // C
        d12, dx = H.highway_gate_bwd(_c(dout), x12, x)
        p = d12.data_ptr()
        H.linear_bwd_data_multi([p, p + C * _F4], 2 * C, [w1, w2], dx, rows, C, accumulate=True)
        dw1 = _emit(w1, lambda out: H.linear_bwd_weight_raw(p, 2 * C, x.data_ptr(), C, out, rows, C, C), (d12, x))
        dw2 = _emit(w2, lambda out: H.linear_bwd_weight_raw(p + C * _F4, 2 * C, x.data_ptr(), C, out, rows, C, C),
                    (d12, x))
        db1 = _emit(b1, lambda out: H.colsum_raw(p, 2 * C, out, rows, C), (d12,), heavy='light')
        db2 = _emit(b2, lambda out: H.colsum_raw(p + C * _F4, 2 * C, out, rows, C), (d12,), heavy='light')
        return dx, dw1, db1, dw2, db2


class HighwayStackFn(Function):
    """A CBHG's stack of HighwayNetworks (common_layers.py:35-40, 86-88, 117-118) with every gate inside a GEMM: the
    forward gate is the epilogue of the layer's one W1 | W2 product (weights interleaved so that a lane holds both
    pre-activations of a unit), and the gate gradient of layer i - 1 is the epilogue of layer i's data-gradient product
    (whose result IS d(out) of layer i - 1).  Only the top layer's gate gradient still runs as a kernel (its d(out)
    comes from the GRU above).  Needs width % 32 == 0; other widths take HighwayFn per layer.
    args: x, then (W1, b1, W2, b2) per layer."""

    @staticmethod
    def forward(ctx, x, *params):
        x = _c(x)
        L = len(params) // 4
        train = any(ctx.needs_input_grad)
        xs, x12s = [x], []
        for i in range(L):
            w1, b1, w2, b2 = params[4 * i:4 * i + 4]
            out, x12 = H.highway_fwd(xs[-1], H.highway_pack(w1, w2), b1, b2, save=train)
            xs.append(out)
            x12s.append(x12)
        if train:
            ctx.save_for_backward(*xs[:L], *x12s, *params)
        ctx.L = L
        return xs[-1]

    @staticmethod
    def backward(ctx, dout):
        L = ctx.L
        saved = ctx.saved_tensors
        xs, x12s, params = saved[:L], saved[L:2 * L], saved[2 * L:]
        C = xs[0].shape[-1]
        rows = xs[0].numel() // C
        d12, dx = H.highway_gate_bwd(_c(dout), x12s[L - 1], xs[L - 1])
        grads = [None] * (4 * L)
        for i in range(L - 1, -1, -1):
            w1, b1, w2, b2 = params[4 * i:4 * i + 4]
            x = xs[i]
            d12_below = H.highway_bwd_data(d12, w1, w2, dx, below=(x12s[i - 1], xs[i - 1]) if i > 0 else None)
            p = d12.data_ptr()
            grads[4 * i] = _emit(w1, lambda out, p=p, x=x: H.linear_bwd_weight_raw(p, 2 * C, x.data_ptr(), C, out, rows, C, C),
                                 (d12, x))
            grads[4 * i + 2] = _emit(w2, lambda out, p=p, x=x: H.linear_bwd_weight_raw(p + C * _F4, 2 * C, x.data_ptr(), C, out,
                                                                                    rows, C, C), (d12, x))
            grads[4 * i + 1] = _emit(b1, lambda out, p=p: H.colsum_raw(p, 2 * C, out, rows, C), (d12,), heavy='light')
            grads[4 * i + 3] = _emit(b2, lambda out, p=p: H.colsum_raw(p + C * _F4, 2 * C, out, rows, C), (d12,),
                                     heavy='light')
            d12 = d12_below
        return (dx, *grads)


def highway_stack(x, highways):
    """x through the HighwayNetwork modules `highways` (each has .W1 / .W2 Linear containers)"""
    if not highways:
        return x
    if x.shape[-1] % 32 == 0 and os.environ.get('FT_HIGHWAY_FUSED', '1') == '1':
        flat = []
        for h in highways:
            flat += [h.W1.weight, h.W1.bias, h.W2.weight, h.W2.bias]
        return HighwayStackFn.apply(x, *flat)
    for h in highways:
        x = HighwayFn.apply(x, h.W1.weight, h.W1.bias, h.W2.weight, h.W2.bias)
    return x


# ---------------------------------------------------------------------------------------------------
def _rnn_param_grads(dxp, dhp, x, hid, G, Hh, params, need_dx, late_last=False):
    """Shared tail of the GRU/LSTM backward: weight / bias / input gradients from the per-step
    pre-activation gradients (dxp wrt input projection, dhp wrt hidden projection).
    x is batch-major [B,T,I]; dxp / dhp / hid are the recurrence's time-major [T,B,*] buffers.
    params = (w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r)."""
    B, T, I = x.shape
    rows = B * T
    GH = G * Hh
    grads: List[Optional[torch.Tensor]] = []
    dx = torch.empty_like(x) if need_dx else None
    dbx = H.colsum(dxp)                       # [2*G*H]
    dbh = dbx if dhp is dxp else H.colsum(dhp)
    if dx is not None:          # both directions' input-projection data gradients in one chained launch
        H.linear_bwd_data_multi([dxp.data_ptr(), dxp.data_ptr() + GH * _F4], 2 * GH, [params[0], params[4]], dx,
                                rows, GH, dy_tm_B=B, dx_tm_B=0)
    gb = _emit_copies([params[2], params[3], params[6], params[7]],
                      [dbx[0:GH], dbh[0:GH], dbx[GH:2 * GH], dbh[GH:2 * GH]])
    for d in range(2):
        w_ih, w_hh = params[4 * d], params[4 * d + 1]
        px = dxp.data_ptr() + d * GH * _F4
        ph = dhp.data_ptr() + d * GH * _F4
        g_ih = _emit(w_ih, lambda out, px=px: H.linear_bwd_weight_raw(
            px, 2 * GH, x.data_ptr(), I, out, rows, I, GH, B=B, T=T, dy_tm=True, x_tm=False), (dxp, x))
        g_hh = _emit(w_hh, lambda out, ph=ph, d=d: H.linear_bwd_weight_raw(
            ph, 2 * GH, hid.data_ptr() + d * Hh * _F4, 2 * Hh, out, rows, Hh, GH, B=B, T=T,
            x_shift=-1 if d == 0 else 1, dy_tm=True, x_tm=True), (dhp, hid), late=late_last and d == 1)
        grads += [g_ih, g_hh, gb[2 * d], gb[2 * d + 1]]
    return dx, grads


class BiGRUFn(Function):
    """nn.GRU(bidirectional=True, batch_first=True), h0=0 (common_layers.py:89,123).
    x is batch-major [B,T,I]; the result (and the gradient it expects) is TIME-major [T,B,2H]."""

    @staticmethod
    def forward(ctx, x, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r):
        x = _c(x)
        Hh = w_hh_f.shape[1]
        train = any(ctx.needs_input_grad)
        nch = H.rnn_overlap(x.shape[0] * x.shape[1], x.shape[1])
        if nch:     # the projection runs in time chunks beside the recurrence (ft_gru_layer_fwd)
            out, gates = H.gru_layer_fwd(x, w_ih_f, w_ih_r, b_ih_f, b_ih_r, w_hh_f, w_hh_r, b_hh_f, b_hh_r, Hh,
                                         save_gates=train, nchunks=nch)
        else:
            xp = H.linear_multi_fwd(x, [w_ih_f, w_ih_r], [b_ih_f, b_ih_r], y_tm_B=x.shape[0])
            out, gates = H.gru_fwd(xp, w_hh_f, w_hh_r, b_hh_f, b_hh_r, Hh, save_gates=train)
        if train:
            ctx.save_for_backward(x, out, gates, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r)
        ctx.Hh = Hh
        return out

    @staticmethod
    def backward(ctx, dout):
        x, out, gates = ctx.saved_tensors[:3]
        params = ctx.saved_tensors[3:]
        Hh = ctx.Hh
        pend = flush_begin()    # queued weight gradients run beside the recurrence, issued right behind its launch
        dxp, dhp = H.gru_bwd(_c(dout), out, gates, H.transpose2d(params[1]), H.transpose2d(params[5]), Hh)
        flush_end(pend)
        dx, g = _rnn_param_grads(dxp, dhp, x, out, 3, Hh, params, ctx.needs_input_grad[0])
        return (dx, *g)


class BiLSTMFn(Function):
    """pack_padded_sequence -> nn.LSTM(bidirectional) -> pad_packed_sequence(padding_value)
    (forward_tacotron.py:147-152); lens=None runs over the padded length (:224)."""

    @staticmethod
    def forward(ctx, x, lens, pad_value, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r):
        x = _c(x)
        Hh = w_hh_f.shape[1]
        train = any(ctx.needs_input_grad)
        nch = H.rnn_overlap(x.shape[0] * x.shape[1], x.shape[1])
        if nch:     # the projection runs in time chunks beside the recurrence (ft_lstm_layer_fwd)
            raw, cst, gates = H.lstm_layer_fwd(x, w_ih_f, w_ih_r, b_ih_f, b_ih_r, w_hh_f, w_hh_r, b_hh_f, b_hh_r, lens,
                                               Hh, save_gates=train, nchunks=nch)
        else:
            xp = H.linear_multi_fwd(x, [w_ih_f, w_ih_r], [b_ih_f, b_ih_r], y_tm_B=x.shape[0])
            raw, cst, gates = H.lstm_fwd(xp, w_hh_f, w_hh_r, b_hh_f, b_hh_r, lens, Hh, save_gates=train)
        if train:
            ctx.save_for_backward(x, raw, cst, gates, lens, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r,
                                  b_ih_r, b_hh_r)
        ctx.Hh = Hh
        ctx.has_lens = lens is not None
        # time-major raw states -> batch-major output, padding_value beyond each item's length
        return H.fill_padded(raw, lens, float(pad_value))

    @staticmethod
    def backward(ctx, dout):
        x, raw, cst, gates, lens = ctx.saved_tensors[:5]
        params = ctx.saved_tensors[5:]
        Hh = ctx.Hh
        dout_tm = H.bt_transpose(_c(dout), True)
        pend = flush_begin()
        dg = H.lstm_bwd(dout_tm, raw, cst, gates, H.transpose2d(params[1]),
                        H.transpose2d(params[5]), lens if ctx.has_lens else None, Hh)
        flush_end(pend)
        # (one of the decoder LSTM's four weight gradients may run at the end of the main stream's chain: GradSink.late)
        dx, g = _rnn_param_grads(dg, dg, x, raw, 4, Hh, params, ctx.needs_input_grad[0], late_last=True)
        return (dx, None, None, *g)


def lr_frames(total: torch.Tensor, pack_lens: Optional[torch.Tensor]) -> int:
    """frames the LengthRegulator's result is produced at (one host sync, like the reference): see LengthRegulateFn"""
    if total.numel() == 0:
        return 0
    if pack_lens is None:
        return int(total.max().item())
    Tm, Lmax = torch.stack([total.max().long(), pack_lens.max().long()]).tolist()
    if Lmax > Tm:
        raise H._lib.FtError(f'LengthRegulator: an item is to be packed with {Lmax} frames but the rounded '
                             f'durations expand to at most {Tm} (the reference fails in its packed LSTM here)')
    return Lmax


class LRBiLSTMFn(Function):
    """LengthRegulator -> pack_padded_sequence -> nn.LSTM(bidirectional) -> pad_packed_sequence as ONE node
    (forward_tacotron.py:145-152, common_layers.py:17-24).

    The LSTM's input projection x W_ih^T + b_ih is a row-wise linear map and the regulator only REPEATS rows, so the two
    commute: the projection is formed once per TOKEN (B x Tx = 4,096 rows at the benchmark shape instead of 26,912
    frames -- 6.6x fewer flops, by the same kernel in the same k order, so every row has exactly the bits the frame-level
    GEMM gives it) and ft_lr_expand_tm writes it out per frame, straight into the recurrence's time-major layout
    (frames beyond an item's length hold the bias: the projection of the regulator's zero rows).  Backward: the
    d(pre-activations) of each token's frames are added up first (ft_lr_bwd_tm -- what the regulator's backward does to
    any gradient), and W_ih's input / weight / bias gradients are token-level GEMMs and sums.  The forward is
    bit-identical to LengthRegulateFn + BiLSTMFn; the backward adds the same numbers in another order (frames of a token
    first), an fp32-class difference.  x is the token-level input [B,Tx,I]; `dur` is clamped in place like every
    LengthRegulator input; lens = the lengths the result is packed with (None: padded length, forward_tacotron.py:224)."""

    @staticmethod
    def forward(ctx, x, dur, lens, pad_value, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r):
        x = _c(x)
        Hh = w_hh_f.shape[1]
        train = any(ctx.needs_input_grad)
        cum, total = H.lr_scan(dur)
        Tm = lr_frames(total, lens)
        # [B,Tx,8H], one row per token, by the kernel the frame-level launch (B * Tm rows) would take
        P = H.linear_multi_fwd(x, [w_ih_f, w_ih_r], [b_ih_f, b_ih_r], as_rows=max(1, x.shape[0] * Tm))
        xp = H.lr_expand_tm(P, cum, Tm, torch.cat([b_ih_f.detach(), b_ih_r.detach()]))
        raw, cst, gates = H.lstm_fwd(xp, w_hh_f, w_hh_r, b_hh_f, b_hh_r, lens, Hh, save_gates=train)
        if train:
            ctx.save_for_backward(x, cum, raw, cst, gates, lens, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r,
                                  b_ih_r, b_hh_r)
        ctx.Hh = Hh
        ctx.has_lens = lens is not None
        return H.fill_padded(raw, lens, float(pad_value))

    @staticmethod
    def backward(ctx, dout):
        x, cum, raw, cst, gates, lens = ctx.saved_tensors[:6]
        params = ctx.saved_tensors[6:]
        Hh = ctx.Hh
        B, Tx, I = x.shape
        T = raw.shape[0]
        GH = 4 * Hh
        dout_tm = H.bt_transpose(_c(dout), True)
        pend = flush_begin()
        dg = H.lstm_bwd(dout_tm, raw, cst, gates, H.transpose2d(params[1]), H.transpose2d(params[5]),
                        lens if ctx.has_lens else None, Hh)
        flush_end(pend)
        dP, dP_all = H.lr_bwd_tm(dg, cum, Tx)                 # [B,Tx,8H]: each token's frames added up
        db = H.colsum(dP_all)                                 # all frames (b_ih and b_hh see the same sum)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            H.linear_bwd_data_multi([dP.data_ptr(), dP.data_ptr() + GH * _F4], 2 * GH, [params[0], params[4]], dx,
                                    B * Tx, GH)
        gb = _emit_copies([params[2], params[3], params[6], params[7]], [db[0:GH], db[0:GH], db[GH:], db[GH:]])
        grads: List[Optional[torch.Tensor]] = []
        for d in range(2):
            w_ih, w_hh = params[4 * d], params[4 * d + 1]
            px = dP.data_ptr() + d * GH * _F4
            ph = dg.data_ptr() + d * GH * _F4
            # (token-level operands, but not the step's tail: they go to the side stream like every frame-side gradient)
            g_ih = _emit(w_ih, lambda out, px=px: H.linear_bwd_weight_raw(px, 2 * GH, x.data_ptr(), I, out, B * Tx, I, GH),
                         (dP, x), inline_ok=False)
            g_hh = _emit(w_hh, lambda out, ph=ph, d=d: H.linear_bwd_weight_raw(
                ph, 2 * GH, raw.data_ptr() + d * Hh * _F4, 2 * Hh, out, B * T, Hh, GH, B=B, T=T,
                x_shift=-1 if d == 0 else 1, dy_tm=True, x_tm=True), (dg, raw), late=d == 1)
            grads += [g_ih, g_hh, gb[2 * d], gb[2 * d + 1]]
        return (dx, None, None, None, *grads)


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
    One host sync reads max_b sum_j r_bj to size the output (the reference syncs here too).

    pack_lens (optional int64 [B], device): the lengths the CALLER is about to pack the result with
    (forward_tacotron.py:147-152).  pad_packed_sequence returns max(pack_lens) frames, so everything downstream of the
    LSTM (lin, postnet BatchNorm statistics, GRU) sees exactly that many; the expansion is therefore produced at
    min(T_lr, max(pack_lens)) frames right away (frames beyond it are never read by a packed LSTM and get no
    gradient).  max(pack_lens) > T_lr makes the reference's LSTM raise; so does this (same sync, no extra one)."""

    @staticmethod
    def forward(ctx, x, dur, pack_lens=None):
        x = _c(x)
        cum, total = H.lr_scan(dur)
        Tm = lr_frames(total, pack_lens)
        ctx.save_for_backward(cum)
        ctx.Tx = x.shape[1]
        return H.lr_expand(x, cum, Tm)

    @staticmethod
    def backward(ctx, dy):
        (cum,) = ctx.saved_tensors
        return H.lr_bwd(_c(dy), cum, ctx.Tx), None, None


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


class ConcatColsFn(Function):
    """torch.cat([a, b2, speaker_emb[:, None, :].repeat(1, T, 1)], dim=2)
    (multi_forward_tacotron.py:39-42,83-85,208-210).  a may be a time-major recurrence output; the speaker
    embedding is input data (no gradient)."""

    @staticmethod
    def forward(ctx, a, b2, semb, B, T, a_time_major):
        a = _c(a)
        ctx.Ca = a.shape[-1]
        ctx.Cb = b2.shape[-1] if b2 is not None else 0
        ctx.a_tm = bool(a_time_major)
        return H.concat_cols(a, _c(b2) if b2 is not None else None, _c(semb) if semb is not None else None, B, T,
                             ctx.a_tm)

    @staticmethod
    def backward(ctx, dout):
        dout = _c(dout)
        da = H.slice_cols(dout, 0, ctx.Ca, dst_time_major=ctx.a_tm) if ctx.needs_input_grad[0] else None
        db = H.slice_cols(dout, ctx.Ca, ctx.Cb) if (ctx.Cb and ctx.needs_input_grad[1]) else None
        return da, db, None, None, None, None


class CrossEntropyFn(Function):
    """nn.CrossEntropyLoss(ignore_index) on logits [..., K] and int64 targets [...]
    (trainer/multi_forward_trainer.py:34,88; the reference transposes to [B,K,T] first, same reduction)."""

    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        logits, target = _c(logits), _c(target)
        loss, inv = H.cross_entropy_fwd(logits, target, int(ignore_index))
        ctx.save_for_backward(logits, target, inv)
        ctx.ignore = int(ignore_index)
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, target, inv = ctx.saved_tensors
        return H.cross_entropy_bwd(logits, target, inv, _c(g).reshape(1).float(), ctx.ignore), None, None


def cross_entropy(logits, target, ignore_index=0):
    return CrossEntropyFn.apply(logits, target, ignore_index)
