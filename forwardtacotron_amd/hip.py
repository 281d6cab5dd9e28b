"""Thin tensor-level wrappers over the C ABI (include/fwdtaco_hip.h).

torch is used here only as the owner of device memory and of the HIP stream; every computation is a
hand-written gfx950 kernel behind libfwdtaco_hip.so.  All functions require contiguous fp32 CUDA(HIP)
tensors and raise if handed anything else -- there is no fallback path.
"""
import contextlib
import ctypes
import os
from typing import List, Optional, Sequence

import torch

from . import _lib

c_void_p = ctypes.c_void_p


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_cur_device = getattr(torch._C, '_cuda_getDevice', None)


def _stream():
    """hipStream_t of torch's current stream on the current device.  Goes straight to the C binding: the public
    `torch.cuda.current_stream()` resolves the device through `torch.cuda.is_available()`, i.e. a device-count
    query (~18 us on this stack) per call -- at ~300 launches per step that alone made the FastPitch step host-bound."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, name: str, dtype=torch.float32):
    if not t.is_cuda:
        raise _lib.FtError(f'{name}: expected a device tensor (the HIP path has no CPU fallback)')
    if t.dtype != dtype:
        raise _lib.FtError(f'{name}: expected {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise _lib.FtError(f'{name}: expected a contiguous tensor')
    return t


_workspaces = {}


def workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch buffer per (device, stream): kernels of one stream run in order, so they can share
    it; concurrent streams (the predictors' side stream) get their own."""
    key = (torch.device(device).index or 0, _stream())
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


def _ptr_array(ts: Sequence[Optional[torch.Tensor]]):
    arr = (ctypes.c_void_p * len(ts))(*[_p(t) for t in ts])
    return arr


# ---------------------------------------------------------------------------------------------------
# matmul precision (process-wide switch of the library, include/fwdtaco_hip.h: ft_set_gemm_precision)
# ---------------------------------------------------------------------------------------------------
_PRECISIONS = {'fp32': 0, 'bf16': 1}
_precision_now = 'fp32'


def gemm_precision_mode() -> str:
    """the mode set_gemm_precision last installed"""
    return _precision_now


def set_gemm_precision(mode: str) -> str:
    """'fp32' (default: fp32-exact products) or 'bf16' (operands rounded to bf16, one bf16 MFMA per product, fp32
    accumulation and outputs); returns the previous mode"""
    if mode not in _PRECISIONS:
        raise _lib.FtError(f"gemm precision must be 'fp32' or 'bf16', got {mode!r}")
    global _precision_now
    old = _lib.lib().ft_set_gemm_precision(_PRECISIONS[mode])
    _precision_now = mode
    return 'bf16' if old else 'fp32'


@contextlib.contextmanager
def gemm_precision(mode: str):
    old = set_gemm_precision(mode)
    try:
        yield
    finally:
        set_gemm_precision(old)


# ---------------------------------------------------------------------------------------------------
# fused attention (bf16 mode)
# ---------------------------------------------------------------------------------------------------
def attn_fwd(qkv: torch.Tensor, key_pad: Optional[torch.Tensor], nheads: int, scale: float, p_drop: float, seed: int):
    """qkv [B,T,3d] -> (att [B,T,d], lse2 [B,nheads,T]); include/fwdtaco_hip.h: ft_attn_fwd"""
    _chk(qkv, 'qkv')
    B, T, d3 = qkv.shape
    d = d3 // 3
    att = torch.empty(B, T, d, device=qkv.device, dtype=qkv.dtype)
    lse2 = torch.empty(B, nheads, T, device=qkv.device, dtype=qkv.dtype)
    _lib.call('ft_attn_fwd', _p(qkv), _p(key_pad), _p(att), _p(lse2), B, T, nheads, d // nheads, float(scale),
              float(p_drop), int(seed) & 0xFFFFFFFFFFFFFFFF, _stream())
    return att, lse2


def attn_bwd(qkv, att, datt, key_pad, lse2, nheads: int, scale: float, p_drop: float, seed: int) -> torch.Tensor:
    _chk(datt, 'datt')
    B, T, d3 = qkv.shape
    d = d3 // 3
    dqkv = torch.empty_like(qkv)
    ws = workspace(_lib.query('ft_attn_workspace', B, T, nheads), qkv.device)
    _lib.call('ft_attn_bwd', _p(qkv), _p(att), _p(datt), _p(key_pad), _p(lse2), _p(dqkv), B, T, nheads, d // nheads,
              float(scale), float(p_drop), int(seed) & 0xFFFFFFFFFFFFFFFF, _p(ws), ws.numel(), _stream())
    return dqkv


# ---------------------------------------------------------------------------------------------------
# linear
# ---------------------------------------------------------------------------------------------------
def linear_fwd(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, relu: bool = False,
               out: Optional[torch.Tensor] = None, x_tm_B: int = 0, y_tm_B: int = 0) -> torch.Tensor:
    """x [..., in] , w [out, in] -> [..., out].  x_tm_B / y_tm_B > 0: that side is time-major [T,B,*]
    (see include/fwdtaco_hip.h, "Row layouts"); a layout change swaps the two leading dims of the result."""
    _chk(x, 'x'); _chk(w, 'w')
    in_f = x.shape[-1]
    rows = x.numel() // max(in_f, 1) if in_f else 0
    out_f = w.shape[0]
    assert w.shape[1] == in_f
    if out is not None:
        y = out
    elif bool(x_tm_B) != bool(y_tm_B):
        Bq = x_tm_B or y_tm_B
        lead = (rows // Bq, Bq) if y_tm_B else (Bq, rows // Bq)
        y = torch.empty(*lead, out_f, device=x.device, dtype=x.dtype)
    else:
        y = torch.empty(*x.shape[:-1], out_f, device=x.device, dtype=x.dtype)
    _lib.call('ft_linear_fwd', _p(x), in_f, _p(w), _p(bias), _p(y), out_f, rows, in_f, out_f, int(relu), 0,
              x_tm_B, y_tm_B, _stream())
    return y


def linear_multi_fwd(x: torch.Tensor, ws: List[torch.Tensor], biases: Optional[List[Optional[torch.Tensor]]],
                     relu: bool = False, y_tm_B: int = 0, as_rows: int = 0) -> torch.Tensor:
    """Several Linear layers on one (batch-major) input, outputs concatenated along the last dim (one launch).
    y_tm_B > 0: x is [B,T,in] and the result is written time-major [T,B,sum(out)]."""
    _chk(x, 'x')
    in_f = x.shape[-1]
    rows = x.numel() // max(in_f, 1)
    outs = [int(w.shape[0]) for w in ws]
    offs = [sum(outs[:i]) for i in range(len(outs))]
    lead = (rows // y_tm_B, y_tm_B) if y_tm_B else tuple(x.shape[:-1])
    y = torch.empty(*lead, sum(outs), device=x.device, dtype=x.dtype)
    n = len(ws)
    wa = _ptr_array(ws)
    ba = _ptr_array(biases) if biases is not None else None
    oa = (ctypes.c_int * n)(*offs)
    fa = (ctypes.c_int * n)(*outs)
    if as_rows:         # rounded like a launch over as_rows rows (ft_linear_multi_fwd_as)
        assert not relu and not y_tm_B
        _lib.call('ft_linear_multi_fwd_as', _p(x), in_f, n, ctypes.cast(wa, c_void_p),
                  ctypes.cast(ba, c_void_p) if ba is not None else None, _p(y), sum(outs),
                  ctypes.cast(oa, c_void_p), ctypes.cast(fa, c_void_p), rows, in_f, max(int(as_rows), 1), _stream())
        return y
    _lib.call('ft_linear_multi_fwd', _p(x), in_f, n, ctypes.cast(wa, c_void_p),
              ctypes.cast(ba, c_void_p) if ba is not None else None, _p(y), sum(outs),
              ctypes.cast(oa, c_void_p), ctypes.cast(fa, c_void_p), rows, in_f, int(relu), 0, y_tm_B, _stream())
    return y


# Data gradients contract over the OUTPUT features, which are the slow index of torch's [out,in] weights.  The
# bf16-split MFMA kernel wants the contraction index contiguous in both operands, so the (small) weight is transposed
# first and the product runs in the NT form; NT_GRADS = False keeps the [K][N] form on the f32 MFMA kernel.
NT_GRADS = True


def _wt(w: torch.Tensor):
    """(pointer-holder tensor, flag) of the weight operand of a data gradient"""
    if NT_GRADS and w.shape[0] % 4 == 0:
        return transpose2d(w), 1
    return w, 0


def linear_bwd_data(dy: torch.Tensor, w: torch.Tensor, dx: Optional[torch.Tensor] = None, accumulate: bool = False,
                    dy_tm_B: int = 0, dx_tm_B: int = 0) -> torch.Tensor:
    """dx = dy @ w ; *_tm_B > 0: that side is time-major."""
    _chk(w, 'w'); _chk(dy, 'dy')
    out_f = dy.shape[-1]
    in_f = w.shape[1]
    rows = dy.numel() // max(out_f, 1)
    if dx is None:
        if bool(dy_tm_B) != bool(dx_tm_B):
            Bq = dy_tm_B or dx_tm_B
            lead = (rows // Bq, Bq) if dx_tm_B else (Bq, rows // Bq)
        else:
            lead = tuple(dy.shape[:-1])
        dx = torch.empty(*lead, in_f, device=dy.device, dtype=dy.dtype)
    wt, flag = _wt(w)
    _lib.call('ft_linear_bwd_data', _p(dy), out_f, _p(wt), _p(dx), in_f, rows, in_f, out_f, int(accumulate),
              dy_tm_B, dx_tm_B, flag, _stream())
    return dx


def linear_bwd_data_raw(dy_ptr: int, lddy: int, w: torch.Tensor, dx: torch.Tensor, rows: int, out_f: int,
                        accumulate: bool, dy_tm_B: int = 0, dx_tm_B: int = 0) -> None:
    in_f = w.shape[1]
    wt, flag = _wt(w)
    _lib.call('ft_linear_bwd_data', dy_ptr, lddy, _p(wt), _p(dx), in_f, rows, in_f, out_f, int(accumulate),
              dy_tm_B, dx_tm_B, flag, _stream())


def linear_bwd_data_multi(dy_ptrs: Sequence[int], lddy: int, ws: Sequence[torch.Tensor], dx: torch.Tensor, rows: int,
                          out_f: int, accumulate: bool = False, dy_tm_B: int = 0, dx_tm_B: int = 0) -> None:
    """dx (+)= sum_i dy_i @ w_i in one chained launch (dy_i given as raw device addresses, common row stride)"""
    n = len(ws)
    in_f = ws[0].shape[1]
    da = (ctypes.c_void_p * n)(*[int(p) for p in dy_ptrs])
    flag = 1 if NT_GRADS and all(w.shape[0] % 4 == 0 for w in ws) else 0
    wts = [transpose2d(w) for w in ws] if flag else list(ws)
    wa = _ptr_array(wts)
    nbytes = _lib.lib().ft_linear_bwd_data_multi_workspace(n, rows, in_f, out_f, _stream()) if flag else 0
    if nbytes:          # few output tiles, long contraction: split-K (scratch from the stream's workspace)
        ws = workspace(nbytes, dx.device)
        _lib.call('ft_linear_bwd_data_multi_ws', n, ctypes.cast(da, c_void_p), lddy, ctypes.cast(wa, c_void_p), _p(dx),
                  in_f, rows, in_f, out_f, int(accumulate), dy_tm_B, dx_tm_B, flag, _p(ws), ws.numel(), _stream())
        return
    _lib.call('ft_linear_bwd_data_multi', n, ctypes.cast(da, c_void_p), lddy, ctypes.cast(wa, c_void_p), _p(dx), in_f,
              rows, in_f, out_f, int(accumulate), dy_tm_B, dx_tm_B, flag, _stream())


def linear_bwd_weight_raw(dy_ptr: int, lddy: int, x_ptr: int, ldx: int, dw: torch.Tensor, rows: int, in_f: int,
                          out_f: int, B: int = 1, T: int = 0, x_shift: int = 0, accumulate: bool = False,
                          dy_tm: bool = False, x_tm: bool = False) -> None:
    nbytes = _lib.query('ft_linear_bwd_weight_workspace', rows, in_f, out_f)
    ws = workspace(nbytes, dw.device)
    _lib.call('ft_linear_bwd_weight', dy_ptr, lddy, x_ptr, ldx, _p(dw), rows, in_f, out_f, B, T if T else rows,
              x_shift, int(accumulate), int(dy_tm), int(x_tm), _p(ws), ws.numel(), _stream())


def linear_bwd_weight(dy: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """dw[out,in] = dy^T x over all leading positions."""
    _chk(dy, 'dy'); _chk(x, 'x')
    out_f, in_f = dy.shape[-1], x.shape[-1]
    rows = x.numel() // max(in_f, 1)
    dw = torch.empty(out_f, in_f, device=x.device, dtype=x.dtype)
    linear_bwd_weight_raw(_p(dy), out_f, _p(x), in_f, dw, rows, in_f, out_f)
    return dw


# ---------------------------------------------------------------------------------------------------
# per-step operand packs
# ---------------------------------------------------------------------------------------------------
class PackCache:
    """Every re-laid-out copy of a model's weights the GEMM kernels consume -- tap-major conv packs [k,Cout,Cin],
    their transposes [k,Cin,Cout], the contiguous per-bank concatenations of both, and W^T of every matrix -- kept in
    persistent buffers and refreshed by ONE ft_pack_weights launch.  Weights change only at the optimizer step, so a
    training step calls refresh() once up front instead of ~120 small per-layer launches spread over forward and
    backward.  Only valid between a refresh() and the next weight update: the owner (trainer.TrainStep) installs it as
    `hip.pack_cache` for the duration of one step and removes it afterwards; with no cache installed every wrapper
    below packs on the fly as before.

    mats: 2-D weights; convs: 3-D Conv1d weights used on their own; banks: lists of Conv1d weights (k = 1..K, equal
    [C,Cin]) whose packs must sit back to back (CBHG conv bank)."""

    def __init__(self, mats, convs, banks, device, highways=()):
        """highways: (W1, W2) pairs of HighwayNetworks whose width is a multiple of 32 -> their 32-row interleave
        (ft_highway_pack layout), the B operand of the fused highway forward"""
        import struct
        self.wp, self.wpt, self.t2d, self.bank, self.hw = {}, {}, {}, {}, {}
        self._keep = []
        descs = []
        tiles = 0

        def add(src, dst, dst_t, d0, d1, k):
            nonlocal tiles
            descs.append(struct.pack('PPPqiiii', src.data_ptr(), _p(dst) or 0, _p(dst_t) or 0, tiles, d0, d1, k, 0))
            tiles += k * ((d0 + 31) // 32) * ((d1 + 31) // 32)

        for w in mats:
            _chk(w, 'w')
            R, C = w.shape
            wt = torch.empty(C, R, device=device, dtype=w.dtype)
            self.t2d[(w.data_ptr(), R, C)] = wt
            add(w, None, wt, R, C, 1)
        for w in convs:
            _chk(w, 'w')
            Cout, Cin, k = w.shape
            wp = torch.empty(k, Cout, Cin, device=device, dtype=w.dtype)
            wpt = torch.empty(k, Cin, Cout, device=device, dtype=w.dtype)
            self.wp[w.data_ptr()] = wp
            self.wpt[w.data_ptr()] = wpt
            add(w, wp, wpt, Cout, Cin, k)
        for ws in banks:
            C, Cin = ws[0].shape[0], ws[0].shape[1]
            n = sum(w.shape[2] for w in ws) * C * Cin
            wp_all = torch.empty(n, device=device, dtype=ws[0].dtype)
            wpt_all = torch.empty(n, device=device, dtype=ws[0].dtype)
            off = 0
            for w in ws:
                _chk(w, 'w')
                k = w.shape[2]
                add(w, wp_all[off:off + k * C * Cin], wpt_all[off:off + k * C * Cin], C, Cin, k)
                off += k * C * Cin
            self.bank[tuple(w.data_ptr() for w in ws)] = (wp_all, wpt_all)
        for w1, w2 in highways:
            _chk(w1, 'w1'); _chk(w2, 'w2')
            C = w1.shape[0]
            if w1.shape != (C, C) or w2.shape != (C, C) or C % 32:
                continue
            pack = torch.empty(2 * C, C, device=device, dtype=w1.dtype)
            self.hw[(w1.data_ptr(), w2.data_ptr())] = pack
            for j in range(C // 32):        # plain [32, C] row-block copies (dst only, one tap)
                add(w1[32 * j:32 * j + 32], pack[64 * j:64 * j + 32], None, 32, C, 1)
                add(w2[32 * j:32 * j + 32], pack[64 * j + 32:64 * j + 64], None, 32, C, 1)
        self.n, self.tiles = len(descs), tiles
        self.descs = torch.frombuffer(bytearray(b''.join(descs)), dtype=torch.uint8).to(device) if descs else None

    def refresh(self) -> None:
        if self.n:
            _lib.call('ft_pack_weights', _p(self.descs), self.n, self.tiles, _stream())

    @staticmethod
    def release() -> None:
        """the weights are about to change (or the step failed): the packs are stale until the next refresh()"""


pack_cache: Optional[PackCache] = None


def bank_packs(ws: Sequence[torch.Tensor], transposed: bool) -> torch.Tensor:
    """back-to-back tap-major packs ([k,C,Cin], or [k,Cin,C] if transposed) of the members of a conv bank"""
    if pack_cache is not None:
        hit = pack_cache.bank.get(tuple(w.data_ptr() for w in ws))
        if hit is not None:
            return hit[1 if transposed else 0]
    C, Cin = ws[0].shape[0], ws[0].shape[1]
    out = torch.empty(sum(w.shape[2] for w in ws) * C * Cin, device=ws[0].device, dtype=ws[0].dtype)
    off = 0
    for w in ws:
        k = w.shape[2]
        n = k * C * Cin
        if transposed:
            conv_pack_weight_t(w, out=out[off:off + n].view(k, Cin, C))
        else:
            conv_pack_weight(w, out=out[off:off + n].view(k, C, Cin))
        off += n
    return out


# ---------------------------------------------------------------------------------------------------
# channels-last conv
# ---------------------------------------------------------------------------------------------------
def conv_pack_weight(w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[Cout,Cin,k] -> tap-major [k,Cout,Cin]"""
    if out is None and pack_cache is not None:
        hit = pack_cache.wp.get(w.data_ptr())
        if hit is not None:
            return hit
    _chk(w, 'w')
    Cout, Cin, k = w.shape
    wp = out if out is not None else torch.empty(k, Cout, Cin, device=w.device, dtype=w.dtype)
    _lib.call('ft_conv_pack_weight', _p(w), _p(wp), Cout, Cin, k, _stream())
    return wp


def conv1d_fwd(x: torch.Tensor, wp: torch.Tensor, relu: bool, Tout: Optional[int] = None,
               scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None,
               accumulate_into: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [B,T,Cin], wp [k,Cout,Cin] -> y [B,Tout,Cout]; accumulate_into: result is ADDED onto that tensor"""
    _chk(x, 'x'); _chk(wp, 'wp')
    B, T, Cin = x.shape
    k, Cout, _ = wp.shape
    Tout = T if Tout is None else Tout
    y = accumulate_into if accumulate_into is not None else torch.empty(B, Tout, Cout, device=x.device,
                                                                        dtype=x.dtype)
    _lib.call('ft_conv1d_fwd', _p(x), Cin, _p(wp), _p(scale), _p(shift), _p(y), Cout, B, T, Cin, Cout, k, Tout,
              int(relu), int(accumulate_into is not None), _stream())
    return y


def conv_bank_fwd(x: torch.Tensor, wp_all: torch.Tensor, K: int, C: int, relu: bool, Tout: int,
                  scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [B,T,Cin]; wp_all flat packed weights of members k=1..K -> ybank [B,Tout,K*C]"""
    _chk(x, 'x'); _chk(wp_all, 'wp_all')
    B, T, Cin = x.shape
    assert wp_all.numel() == C * Cin * K * (K + 1) // 2
    y = torch.empty(B, Tout, K * C, device=x.device, dtype=x.dtype)
    _lib.call('ft_conv_bank_fwd', _p(x), Cin, _p(wp_all), _p(scale), _p(shift), _p(y), B, T, Cin, C, K, Tout,
              int(relu), _stream())
    return y


def conv1d_fwd_stats(x: torch.Tensor, wp: torch.Tensor, relu: bool, Tout: int):
    """conv (+ReLU) whose GEMM epilogue also leaves the BatchNorm statistics partials of y behind
    (include/fwdtaco_hip.h: ft_conv1d_fwd_stats).  -> (y [B,Tout,Cout], partial buffer, nchunks)"""
    _chk(x, 'x'); _chk(wp, 'wp')
    B, T, Cin = x.shape
    k, Cout, _ = wp.shape
    y = torch.empty(B, Tout, Cout, device=x.device, dtype=x.dtype)
    nbytes = _lib.query('ft_conv1d_fwd_stats_workspace', B, Tout, Cout, k)
    part = workspace(nbytes, x.device)
    n = ctypes.c_int(0)
    _lib.call('ft_conv1d_fwd_stats', _p(x), Cin, _p(wp), _p(y), Cout, B, T, Cin, Cout, k, Tout, int(relu), _p(part),
              part.numel(), ctypes.byref(n), _stream())
    return y, part, n.value


def conv_bank_fwd_stats(x: torch.Tensor, wp_all: torch.Tensor, K: int, C: int, relu: bool):
    """training-mode conv bank: ybank [B,T+1,K*C] + the statistics partials of its K BatchNorms"""
    _chk(x, 'x'); _chk(wp_all, 'wp_all')
    B, T, Cin = x.shape
    assert wp_all.numel() == C * Cin * K * (K + 1) // 2
    y = torch.empty(B, T + 1, K * C, device=x.device, dtype=x.dtype)
    nbytes = _lib.query('ft_conv_stats_workspace', B, T + 1, K * C)
    part = workspace(nbytes, x.device)
    n = ctypes.c_int(0)
    _lib.call('ft_conv_bank_fwd_stats', _p(x), Cin, _p(wp_all), _p(y), B, T, Cin, C, K, int(relu), _p(part),
              part.numel(), ctypes.byref(n), _stream())
    return y, part, n.value


def bn_train_from_partials(part: torch.Tensor, nchunks: int, y: torch.Tensor, gamma, beta, running_mean, running_var,
                           Tout: int, group: int = 0, residual: Optional[torch.Tensor] = None,
                           momentum: float = None, eps: float = None):
    """finalize the statistics partials (ordered) + normalise: -> (out [B,Tout,C], save_mean, save_rstd)"""
    B, Tbuf, C = y.shape
    out = torch.empty(B, Tout, C, device=y.device, dtype=y.dtype)
    mean = torch.empty(C, device=y.device, dtype=y.dtype)
    rstd = torch.empty(C, device=y.device, dtype=y.dtype)
    _lib.call('ft_bn_train_from_partials', _p(part), nchunks, _p(y), _p(gamma), _p(beta), _p(residual), _p(out),
              _p(running_mean), _p(running_var), None, _p(mean), _p(rstd), B, Tbuf, Tout, C, group,
              BN_MOMENTUM if momentum is None else momentum, BN_EPS if eps is None else eps, _stream())
    return out, mean, rstd


def conv_pack_weight_t(w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[Cout,Cin,k] -> transposed tap-major [k,Cin,Cout] (weight operand of the data gradient in the NT form)"""
    if out is None and pack_cache is not None:
        hit = pack_cache.wpt.get(w.data_ptr())
        if hit is not None:
            return hit
    _chk(w, 'w')
    Cout, Cin, k = w.shape
    wpt = out if out is not None else torch.empty(k, Cin, Cout, device=w.device, dtype=w.dtype)
    _lib.call('ft_conv_pack_weight_t', _p(w), _p(wpt), Cout, Cin, k, _stream())
    return wpt


def conv1d_bwd_data_raw(dy_ptr: int, lddy: int, wp: torch.Tensor, dx: torch.Tensor, B: int, T: int, Tbuf: int,
                        Tvalid: int, accumulate: bool, w: Optional[torch.Tensor] = None) -> None:
    """wp: packed [k,Cout,Cin]; with the original weight `w` at hand (and Cout % 4 == 0) the NT form is used"""
    k, Cout, Cin = wp.shape
    if NT_GRADS and w is not None and Cout % 4 == 0:
        wpt = conv_pack_weight_t(w)
        _lib.call('ft_conv1d_bwd_data', dy_ptr, lddy, _p(wpt), _p(dx), Cin, B, T, Cin, Cout, k, Tbuf, Tvalid,
                  int(accumulate), 1, _stream())
        return
    _lib.call('ft_conv1d_bwd_data', dy_ptr, lddy, _p(wp), _p(dx), Cin, B, T, Cin, Cout, k, Tbuf, Tvalid,
              int(accumulate), 0, _stream())


def conv_bank_bwd_data(dy: torch.Tensor, wp_all: torch.Tensor, K: int, C: int, Cin: int, T: int,
                       ws: Optional[Sequence[torch.Tensor]] = None) -> torch.Tensor:
    """dy [B,Tbuf,K*C] (gradient of the bank buffer) -> dx [B,T,Cin], all K members in one chained launch.
    ws: the members' original [C,Cin,k] weights -> transposed packs, NT form."""
    _chk(dy, 'dy'); _chk(wp_all, 'wp_all')
    B, Tbuf, _ = dy.shape
    dx = torch.empty(B, T, Cin, device=dy.device, dtype=dy.dtype)
    flag = 0
    if NT_GRADS and ws is not None and C % 4 == 0:
        wp_all, flag = bank_packs(ws, True), 1
    nbytes = _lib.query('ft_conv_bank_bwd_data_workspace', B, T, Cin, K)
    ws = workspace(nbytes, dy.device) if nbytes else None
    _lib.call('ft_conv_bank_bwd_data', _p(dy), K * C, _p(wp_all), _p(dx), Cin, B, T, Cin, C, K, Tbuf, flag,
              _p(ws), ws.numel() if ws is not None else 0, _stream())
    return dx


def conv_bank_bwd_weight(dy: torch.Tensor, x: torch.Tensor, dws: Sequence[torch.Tensor], C: int) -> None:
    """dy [B,Tbuf,K*C] gradient of the bank buffer, x [B,T,Cin]; overwrites dws[i] ([C,Cin,i+1]) for all K members
    with one GEMM launch (C % 128 == 0)."""
    _chk(dy, 'dy'); _chk(x, 'x')
    B, T, Cin = x.shape
    Tbuf, K = dy.shape[1], len(dws)
    for i, d in enumerate(dws):
        _chk(d, 'dw')
        if tuple(d.shape) != (C, Cin, i + 1):
            raise _lib.FtError(f'conv_bank_bwd_weight: dw[{i}] has shape {tuple(d.shape)}')
    nbytes = _lib.query('ft_conv_bank_bwd_weight_workspace', B, T, Cin, C, K, Tbuf)
    ws = workspace(nbytes, x.device)
    _lib.call('ft_conv_bank_bwd_weight', _p(dy), K * C, _p(x), Cin, _ptr_array(dws), B, T, Cin, C, K, Tbuf, _p(ws),
              ws.numel(), _stream())


def conv1d_bwd_weight_raw(dy_ptr: int, lddy: int, x: torch.Tensor, dw: torch.Tensor, Tbuf: int, Tvalid: int) -> None:
    B, T, Cin = x.shape
    Cout, _, k = dw.shape
    nbytes = _lib.query('ft_conv1d_bwd_weight_workspace', B, T, Cin, Cout, k, Tvalid)
    ws = workspace(nbytes, x.device)
    _lib.call('ft_conv1d_bwd_weight', dy_ptr, lddy, _p(x), Cin, _p(dw), B, T, Cin, Cout, k, Tbuf, Tvalid, _p(ws),
              ws.numel(), _stream())


# ---------------------------------------------------------------------------------------------------
# LengthRegulator
# ---------------------------------------------------------------------------------------------------
def lr_scan(dur: torch.Tensor):
    """In-place clamp of dur (<0 -> 0) + prefix sums.  Returns (cum int32 [B,Tx+1], total int32 [B])."""
    _chk(dur, 'dur')
    B, Tx = dur.shape
    cum = torch.empty(B, Tx + 1, device=dur.device, dtype=torch.int32)
    total = torch.empty(B, device=dur.device, dtype=torch.int32)
    _lib.call('ft_lr_scan', _p(dur), B, Tx, _p(cum), _p(total), _stream())
    return cum, total


def lr_expand(x: torch.Tensor, cum: torch.Tensor, Tm: int, want_src: bool = False):
    _chk(x, 'x'); _chk(cum, 'cum', torch.int32)
    B, Tx, C = x.shape
    y = torch.empty(B, Tm, C, device=x.device, dtype=x.dtype)
    src = torch.empty(B, Tm, device=x.device, dtype=torch.int32) if want_src else None
    _lib.call('ft_lr_expand', _p(x), _p(cum), _p(y), _p(src), B, Tx, Tm, C, _stream())
    return (y, src) if want_src else y


def lr_bwd(dy: torch.Tensor, cum: torch.Tensor, Tx: int) -> torch.Tensor:
    _chk(dy, 'dy'); _chk(cum, 'cum', torch.int32)
    B, Tm, C = dy.shape
    dx = torch.empty(B, Tx, C, device=dy.device, dtype=dy.dtype)
    _lib.call('ft_lr_bwd', _p(dy), _p(cum), _p(dx), B, Tx, Tm, C, _stream())
    return dx


def lr_expand_tm(x: torch.Tensor, cum: torch.Tensor, Tm: int, pad_row: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [B,Tx,C] -> TIME-major [Tm,B,C]; frames beyond an item's length hold pad_row (zeros if None)"""
    _chk(x, 'x'); _chk(cum, 'cum', torch.int32)
    B, Tx, C = x.shape
    if pad_row is not None:
        _chk(pad_row, 'pad_row')
        assert pad_row.numel() == C
    y = torch.empty(Tm, B, C, device=x.device, dtype=x.dtype)
    _lib.call('ft_lr_expand_tm', _p(x), _p(cum), _p(pad_row), _p(y), B, Tx, Tm, C, _stream())
    return y


def lr_bwd_tm(dy: torch.Tensor, cum: torch.Tensor, Tx: int):
    """dy TIME-major [Tm,B,C] -> (dx [B,Tx,C]: the frames of every token added up, in frame order; rows [B*Tx + B, C]:
    dx's rows followed by one row per item with the sum of the frames beyond its last token -- what a column sum over
    ALL frames has to run over)"""
    _chk(dy, 'dy'); _chk(cum, 'cum', torch.int32)
    Tm, B, C = dy.shape
    rows = torch.empty(B * Tx + B, C, device=dy.device, dtype=dy.dtype)
    dx = rows[:B * Tx].view(B, Tx, C)
    _lib.call('ft_lr_bwd_tm', _p(dy), _p(cum), _p(dx), _p(rows[B * Tx:]), B, Tx, Tm, C, _stream())
    return dx, rows


# ---------------------------------------------------------------------------------------------------
# BatchNorm / column sums
# ---------------------------------------------------------------------------------------------------
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def bn_train_fwd(y: torch.Tensor, gamma, beta, running_mean, running_var, Tout: int, group: int = 0,
                 residual: Optional[torch.Tensor] = None, momentum: float = BN_MOMENTUM, eps: float = BN_EPS):
    """y [B,Tbuf,C] -> (out [B,Tout,C], save_mean, save_rstd); running stats updated in place."""
    _chk(y, 'y')
    B, Tbuf, C = y.shape
    out = torch.empty(B, Tout, C, device=y.device, dtype=y.dtype)
    mean = torch.empty(C, device=y.device, dtype=y.dtype)
    rstd = torch.empty(C, device=y.device, dtype=y.dtype)
    nbytes = _lib.query('ft_bn_workspace', B, Tbuf, C)
    ws = workspace(nbytes, y.device)
    _lib.call('ft_bn_train_fwd', _p(y), _p(gamma), _p(beta), _p(residual), _p(out), _p(running_mean),
              _p(running_var), None, _p(mean), _p(rstd), B, Tbuf, Tout, C, group, momentum, eps, _p(ws), ws.numel(),
              _stream())
    return out, mean, rstd


def bn_bwd(dout: torch.Tensor, y: torch.Tensor, gamma, mean, rstd, group: int, relu: bool,
           dgamma: Optional[torch.Tensor] = None, dbeta: Optional[torch.Tensor] = None):
    """-> (dy [B,Tbuf,C], dgamma [C], dbeta [C]); dgamma / dbeta may be caller-provided output buffers"""
    _chk(dout, 'dout'); _chk(y, 'y')
    B, Tbuf, C = y.shape
    Tout = dout.shape[1]
    dy = torch.empty_like(y)
    if dgamma is None:
        dgamma = torch.empty(C, device=y.device, dtype=y.dtype)
    if dbeta is None:
        dbeta = torch.empty(C, device=y.device, dtype=y.dtype)
    nbytes = _lib.query('ft_bn_workspace', B, Tbuf, C)
    ws = workspace(nbytes, y.device)
    _lib.call('ft_bn_bwd', _p(dout), _p(y), _p(gamma), _p(mean), _p(rstd), _p(dy), _p(dgamma), _p(dbeta), B, Tbuf,
              Tout, C, group, int(relu), _p(ws), ws.numel(), _stream())
    return dy, dgamma, dbeta


def bn_pool_fusable(y: torch.Tensor, group: int) -> bool:
    """the fused BatchNorm + MaxPool kernels take the 16-byte path only"""
    return y.shape[-1] % 4 == 0 and group % 4 == 0 and y.data_ptr() % 16 == 0


def bn_pool_from_partials(part: torch.Tensor, nchunks: int, y: torch.Tensor, gamma, beta, running_mean, running_var,
                          Tout: int, group: int = 0, momentum: float = None, eps: float = None):
    """finalize the statistics partials + MaxPool1d(2,1,1)(bn(y))[:Tout] in one pass over y (the normalised tensor is
    never stored): -> (out [B,Tout,C], save_mean, save_rstd)"""
    B, Tbuf, C = y.shape
    out = torch.empty(B, Tout, C, device=y.device, dtype=y.dtype)
    mean = torch.empty(C, device=y.device, dtype=y.dtype)
    rstd = torch.empty(C, device=y.device, dtype=y.dtype)
    _lib.call('ft_bn_pool_from_partials', _p(part), nchunks, _p(y), _p(gamma), _p(beta), _p(out), _p(running_mean),
              _p(running_var), None, _p(mean), _p(rstd), B, Tbuf, Tout, C, group,
              BN_MOMENTUM if momentum is None else momentum, BN_EPS if eps is None else eps, _stream())
    return out, mean, rstd


def bn_pool_bwd(dout: torch.Tensor, y: torch.Tensor, gamma, beta, mean, rstd, group: int, relu: bool):
    """backward of bn_pool_from_partials: dout = gradient of the pooled output -> (dy [B,Tbuf,C], dgamma, dbeta)"""
    _chk(dout, 'dout'); _chk(y, 'y')
    B, Tbuf, C = y.shape
    Tout = dout.shape[1]
    dy = torch.empty_like(y)
    dgamma = torch.empty(C, device=y.device, dtype=y.dtype)
    dbeta = torch.empty(C, device=y.device, dtype=y.dtype)
    nbytes = _lib.query('ft_bn_workspace', B, Tbuf, C)
    ws = workspace(nbytes, y.device)
    _lib.call('ft_bn_pool_bwd', _p(dout), _p(y), _p(gamma), _p(beta), _p(mean), _p(rstd), _p(dy), _p(dgamma), _p(dbeta),
              B, Tbuf, Tout, C, group, int(relu), _p(ws), ws.numel(), _stream())
    return dy, dgamma, dbeta


def bn_fold_eval(gamma, beta, running_mean, running_var, eps: float = BN_EPS):
    C = gamma.numel()
    scale = torch.empty(C, device=gamma.device, dtype=gamma.dtype)
    shift = torch.empty(C, device=gamma.device, dtype=gamma.dtype)
    _lib.call('ft_bn_fold_eval', _p(gamma), _p(beta), _p(running_mean), _p(running_var), eps, _p(scale), _p(shift),
              C, _stream())
    return scale, shift


def colsum_raw(x_ptr: int, ldx: int, out: torch.Tensor, rows: int, C: int, scale: float = 1.0,
               accumulate: bool = False) -> None:
    nbytes = _lib.query('ft_colsum_workspace', rows, C)
    ws = workspace(nbytes, out.device)
    _lib.call('ft_colsum', x_ptr, ldx, _p(out), rows, C, scale, int(accumulate), _p(ws), ws.numel(), _stream())


def colsum2_raw(x0_ptr: int, x1_ptr: int, ldx: int, out0: torch.Tensor, out1: torch.Tensor, rows: int, C: int) -> None:
    """out0 = column sums of x0, out1 = of x1 (same [rows, C] shape, row stride ldx): one partial + one finalize launch"""
    nbytes = _lib.query('ft_colsum_workspace', rows, C)
    ws = workspace(nbytes, out0.device)
    _lib.call('ft_colsum2', x0_ptr, x1_ptr, ldx, _p(out0), _p(out1), rows, C, _p(ws), ws.numel(), _stream())


def colsum(x: torch.Tensor) -> torch.Tensor:
    _chk(x, 'x')
    C = x.shape[-1]
    rows = x.numel() // max(C, 1)
    out = torch.empty(C, device=x.device, dtype=x.dtype)
    colsum_raw(_p(x), C, out, rows, C)
    return out


def colsum_batch(xs: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    """column sums of up to 16 matrices with the same row count in one partial + one finalize launch (each bit-identical
    to colsum of that matrix)"""
    rows = xs[0].numel() // xs[0].shape[-1]
    for x in xs:
        _chk(x, 'x')
        assert x.numel() // x.shape[-1] == rows
    n = len(xs)
    outs = [torch.empty(x.shape[-1], device=x.device, dtype=x.dtype) for x in xs]
    Cs = (ctypes.c_int * n)(*[int(x.shape[-1]) for x in xs])
    lds = (ctypes.c_long * n)(*[int(x.shape[-1]) for x in xs])
    nbytes = _lib.lib().ft_colsum_batch_workspace(n, ctypes.cast(Cs, c_void_p), rows)
    ws = workspace(max(int(nbytes), 16), xs[0].device)
    _lib.call('ft_colsum_batch', n, ctypes.cast(_ptr_array(list(xs)), c_void_p), ctypes.cast(lds, c_void_p),
              ctypes.cast(_ptr_array(outs), c_void_p), ctypes.cast(Cs, c_void_p), rows, _p(ws), ws.numel(), _stream())
    return outs


# ---------------------------------------------------------------------------------------------------
# element-wise
# ---------------------------------------------------------------------------------------------------
def dropout(x: torch.Tensor, p: float, seed: int) -> torch.Tensor:
    _chk(x, 'x')
    out = torch.empty_like(x)
    _lib.call('ft_dropout', _p(x), _p(out), x.numel(), p, seed & 0xFFFFFFFFFFFFFFFF, _stream())
    return out


def scale(x: torch.Tensor, s: float) -> torch.Tensor:
    _chk(x, 'x')
    out = torch.empty_like(x)
    _lib.call('ft_scale', _p(x), _p(out), x.numel(), s, _stream())
    return out



def embedding_fwd(idx: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    _chk(idx, 'idx', torch.int64); _chk(w, 'w')
    V, C = w.shape
    out = torch.empty(*idx.shape, C, device=w.device, dtype=w.dtype)
    err = _err_flag(w.device)
    _lib.call('ft_embedding_fwd', _p(idx), _p(w), _p(out), idx.numel(), C, V, _p(err), _stream())
    return out


_err_flags = {}


def _err_flag(device) -> torch.Tensor:
    key = torch.device(device).index or 0
    if key not in _err_flags:
        _err_flags[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _err_flags[key]


def check_index_errors(device) -> None:
    """Host-side check of the out-of-range embedding index flag (syncs)."""
    f = _err_flag(device)
    if int(f.item()) != 0:
        f.zero_()
        raise IndexError('embedding index out of range (set by ft_embedding_fwd)')


def onehot(idx: torch.Tensor, V: int) -> torch.Tensor:
    _chk(idx, 'idx', torch.int64)
    out = torch.empty(idx.numel(), V, device=idx.device, dtype=torch.float32)
    _lib.call('ft_onehot', _p(idx), _p(out), idx.numel(), V, _stream())
    return out


def embedding_bwd(idx: torch.Tensor, dout: torch.Tensor, V: int,
                  onehot_cache: Optional[dict] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dW[V,C] = onehot(idx)^T @ dout (TN MFMA GEMM).  onehot_cache lets the four embedding tables that
    share one id tensor (main + three predictors) build the one-hot matrix once."""
    _chk(idx, 'idx', torch.int64); _chk(dout, 'dout')
    C = dout.shape[-1]
    key = (idx.data_ptr(), idx.numel(), V, _stream())          # per stream: no cross-stream sharing
    oh = onehot_cache.get(key) if onehot_cache is not None else None
    if oh is None:
        oh = onehot(idx, V)
        if onehot_cache is not None:
            onehot_cache[key] = oh
    dw = out if out is not None else torch.empty(V, C, device=dout.device, dtype=dout.dtype)
    linear_bwd_weight_raw(_p(oh), V, _p(dout), C, dw, idx.numel(), C, V)
    return dw


def highway_gate_fwd(x12: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    _chk(x12, 'x12'); _chk(x, 'x')
    C = x.shape[-1]
    out = torch.empty_like(x)
    _lib.call('ft_highway_gate_fwd', _p(x12), _p(x), _p(out), x.numel() // C, C, _stream())
    return out


def highway_gate_bwd(dout, x12, x):
    _chk(dout, 'dout')
    C = x.shape[-1]
    d12 = torch.empty_like(x12)
    dx = torch.empty_like(x)
    _lib.call('ft_highway_gate_bwd', _p(dout), _p(x12), _p(x), _p(d12), _p(dx), x.numel() // C, C, _stream())
    return d12, dx


def highway_pack(w1: torch.Tensor, w2: torch.Tensor) -> torch.Tensor:
    """the 32-row interleave [2C, C] of W1 / W2 (include/fwdtaco_hip.h: ft_highway_pack); from the step's PackCache if it
    holds this pair"""
    if pack_cache is not None:
        hit = pack_cache.hw.get((w1.data_ptr(), w2.data_ptr()))
        if hit is not None:
            return hit
    _chk(w1, 'w1'); _chk(w2, 'w2')
    C = w1.shape[0]
    out = torch.empty(2 * C, C, device=w1.device, dtype=w1.dtype)
    _lib.call('ft_highway_pack', _p(w1), _p(w2), _p(out), C, _stream())
    return out


def highway_fwd(x: torch.Tensor, w12i: torch.Tensor, b1: torch.Tensor, b2: torch.Tensor, save: bool):
    """HighwayNetwork forward with the gate in the GEMM epilogue -> (out [.., C], x12 [.., 2C] or None)"""
    _chk(x, 'x'); _chk(w12i, 'w12i')
    C = x.shape[-1]
    rows = x.numel() // C
    out = torch.empty_like(x)
    x12 = torch.empty(*x.shape[:-1], 2 * C, device=x.device, dtype=x.dtype) if save else None
    _lib.call('ft_highway_fwd', _p(x), _p(w12i), _p(b1), _p(b2), _p(out), _p(x12), rows, C, _stream())
    return out, x12


def highway_bwd_data(d12: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor, dx: torch.Tensor,
                     below: Optional[tuple] = None) -> Optional[torch.Tensor]:
    """dx += d12[:, :C] W1 + d12[:, C:] W2 (in place; dx holds the direct-path term).  below = (x12, x) of the highway
    layer underneath: dx then becomes THAT layer's direct-path term and its gate gradients d12 [.., 2C] are returned."""
    C = dx.shape[-1]
    rows = dx.numel() // C
    flag = 1 if NT_GRADS else 0
    wa, wb = (transpose2d(w1), transpose2d(w2)) if flag else (w1, w2)
    d12b = torch.empty_like(d12) if below is not None else None
    _lib.call('ft_highway_bwd_data', _p(d12), _p(wa), _p(wb), flag, _p(dx), rows, C,
              _p(below[0]) if below else None, _p(below[1]) if below else None, _p(d12b), _stream())
    return d12b


def maxpool2_fwd(x: torch.Tensor) -> torch.Tensor:
    _chk(x, 'x')
    B, T, C = x.shape
    out = torch.empty_like(x)
    _lib.call('ft_maxpool2_fwd', _p(x), _p(out), B, T, C, _stream())
    return out


def maxpool2_bwd(dout: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    _chk(dout, 'dout')
    B, T, C = x.shape
    dx = torch.empty_like(x)
    _lib.call('ft_maxpool2_bwd', _p(dout), _p(x), _p(dx), B, T, C, _stream())
    return dx


def cond_add_fwd(x, pitch, energy, wp, bp, we, be, sp: float, se: float, x_time_major: bool = False) -> torch.Tensor:
    """x [B,T,C] (or time-major [T,B,C]) -> batch-major [B,T,C]"""
    _chk(x, 'x'); _chk(pitch, 'pitch'); _chk(energy, 'energy')
    B, T = pitch.shape
    C = x.shape[-1]
    out = torch.empty(B, T, C, device=x.device, dtype=x.dtype)
    _lib.call('ft_cond_add_fwd', _p(x), _p(pitch), _p(energy), _p(wp), _p(bp), _p(we), _p(be), sp, se, _p(out), B,
              T, C, int(x_time_major), _stream())
    return out


def cond_taps(pitch, energy) -> torch.Tensor:
    B, T = pitch.shape
    taps = torch.empty(B, T, 8, device=pitch.device, dtype=pitch.dtype)
    _lib.call('ft_cond_taps', _p(pitch), _p(energy), _p(taps), B, T, _stream())
    return taps


def concat_cols(a: torch.Tensor, b2: Optional[torch.Tensor], semb: Optional[torch.Tensor], B: int, T: int,
                a_time_major: bool = False) -> torch.Tensor:
    """[a | b2 | broadcast(semb)] along the feature dim -> batch-major [B,T,Ca+Cb+S]"""
    _chk(a, 'a')
    Ca = a.shape[-1]
    Cb = b2.shape[-1] if b2 is not None else 0
    S = semb.shape[-1] if semb is not None else 0
    out = torch.empty(B, T, Ca + Cb + S, device=a.device, dtype=a.dtype)
    _lib.call('ft_concat_cols', _p(a), Ca, _p(b2), Cb, _p(semb), S, _p(out), B, T, int(a_time_major), _stream())
    return out


def slice_cols(src: torch.Tensor, col0: int, C: int, dst_time_major: bool = False) -> torch.Tensor:
    """src batch-major [B,T,ld] -> src[..., col0:col0+C] as a contiguous [B,T,C] (or time-major [T,B,C])"""
    _chk(src, 'src')
    B, T, ld = src.shape
    dst = torch.empty((T, B, C) if dst_time_major else (B, T, C), device=src.device, dtype=src.dtype)
    _lib.call('ft_slice_cols', _p(src), ld, col0, C, _p(dst), B, T, int(dst_time_major), _stream())
    return dst


def cross_entropy_fwd(logits: torch.Tensor, target: torch.Tensor, ignore_index: int):
    _chk(logits, 'logits'); _chk(target, 'target', torch.int64)
    K = logits.shape[-1]
    rows = logits.numel() // K
    loss = torch.empty((), device=logits.device, dtype=logits.dtype)
    inv = torch.empty(1, device=logits.device, dtype=logits.dtype)
    ws = workspace(_lib.query('ft_cross_entropy_workspace'), logits.device)
    _lib.call('ft_cross_entropy_fwd', _p(logits), _p(target), rows, K, ignore_index, _p(loss), _p(inv), _p(ws),
              ws.numel(), _stream())
    return loss, inv


def cross_entropy_bwd(logits, target, inv, grad_out, ignore_index: int) -> torch.Tensor:
    K = logits.shape[-1]
    rows = logits.numel() // K
    d = torch.empty_like(logits)
    _lib.call('ft_cross_entropy_bwd', _p(logits), _p(target), _p(inv), _p(grad_out), _p(d), rows, K, ignore_index,
              _stream())
    return d


def transpose_pad_fwd(x: torch.Tensor, Tout: int, pad: float) -> torch.Tensor:
    _chk(x, 'x')
    B, T, C = x.shape
    out = torch.empty(B, C, Tout, device=x.device, dtype=x.dtype)
    _lib.call('ft_transpose_pad_fwd', _p(x), _p(out), B, T, C, Tout, pad, _stream())
    return out


def transpose_pad_bwd(dout: torch.Tensor, T: int) -> torch.Tensor:
    _chk(dout, 'dout')
    B, C, Tout = dout.shape
    dx = torch.empty(B, T, C, device=dout.device, dtype=dout.dtype)
    _lib.call('ft_transpose_pad_bwd', _p(dout), _p(dx), B, T, C, Tout, _stream())
    return dx


def copy_segments(srcs: Sequence[torch.Tensor], dsts: Sequence[torch.Tensor]) -> None:
    """dsts[i] <- srcs[i] (contiguous fp32, equal numel), all in one launch (at most 64 pairs per launch)."""
    for i in range(0, len(srcs), 64):
        ss, dd = srcs[i:i + 64], dsts[i:i + 64]
        n = len(ss)
        for a, b in zip(ss, dd):
            _chk(a, 'src'); _chk(b, 'dst')
            if a.numel() != b.numel():
                raise _lib.FtError('copy_segments: size mismatch')
        lens = (ctypes.c_long * n)(*[a.numel() for a in ss])
        _lib.call('ft_copy_segments', _ptr_array(ss), _ptr_array(dd), lens, n, _stream())


def transpose2d(x: torch.Tensor) -> torch.Tensor:
    """[R,C] -> [C,R] (used for W_hh^T in BPTT)."""
    R, C = x.shape
    if pack_cache is not None:
        hit = pack_cache.t2d.get((x.data_ptr(), R, C))
        if hit is not None:
            return hit
    return transpose_pad_fwd(x.view(1, R, C), R, 0.0).view(C, R)


def masked_l1_fwd(x: torch.Tensor, target: torch.Tensor, lens: torch.Tensor):
    _chk(x, 'x'); _chk(target, 'target'); _chk(lens, 'lens', torch.int64)
    B, C, T = x.shape
    loss = torch.empty((), device=x.device, dtype=x.dtype)
    inv = torch.empty(1, device=x.device, dtype=x.dtype)
    ws = workspace(_lib.query('ft_masked_l1_workspace'), x.device)
    _lib.call('ft_masked_l1_fwd', _p(x), _p(target), _p(lens), _p(loss), _p(inv), B, C, T, _p(ws), ws.numel(),
              _stream())
    return loss, inv


def masked_l1_bwd(x, target, lens, inv, grad_out: Optional[torch.Tensor], factor: float = 1.0) -> torch.Tensor:
    B, C, T = x.shape
    dx = torch.empty_like(x)
    _lib.call('ft_masked_l1_bwd', _p(x), _p(target), _p(lens), _p(inv), _p(grad_out), factor, _p(dx), B, C, T,
              _stream())
    return dx


# ---------------------------------------------------------------------------------------------------
# recurrences
# ---------------------------------------------------------------------------------------------------
# All recurrence buffers are TIME-major: xp [T,B,2*G*H], out / cstate [T,B,2H], gates [T,B,2,4H].
_rnn_ws = {}          # (device, stream, gates, B, H) -> workspace of the persistent recurrence kernels


def _rnn_workspace(gates: int, B: int, H: int, device):
    nbytes = _lib.query('ft_rnn_workspace', gates, B, H)
    if nbytes == 0:
        return None, 0
    key = (torch.device(device).index or 0, _stream(), gates, B, H)
    ws = _rnn_ws.get(key)
    if ws is None:
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        _rnn_ws[key] = ws
    return ws, nbytes


def check_rnn_status(clear: bool = True) -> None:
    """Synchronises the device and raises if ANY persistent recurrence launched on it since the fault word was last
    cleared timed out (the word is sticky: later launches, successful or not, never reset it).  clear=True resets it, so
    that a caller who switches to the per-step kernels is not told about the same timeout again.  Training does not
    depend on anyone calling this: the optimizer kernels read the same word on the device and skip the update
    (trainer.TrainStep)."""
    _lib.call('ft_rnn_status', int(clear))


def rnn_note_join(waiting: 'torch.cuda.Stream', joined: 'torch.cuda.Stream') -> None:
    """call right after waiting.wait_stream(joined): see ft_rnn_note_join (include/fwdtaco_hip.h)"""
    _lib.call('ft_rnn_note_join', waiting.cuda_stream, joined.cuda_stream)


def rnn_mode_counts():
    """((direction, batch group) groups that ran XCD-local, groups on the agent-scope protocol) since load; syncs"""
    a, b = ctypes.c_long(0), ctypes.c_long(0)
    _lib.call('ft_rnn_mode_counts', ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value


def rnn_counters():
    """(launches that ran in the persistent form, launches that cannot fit the chip -> per-step kernels) since load"""
    a, b = ctypes.c_long(0), ctypes.c_long(0)
    _lib.call('ft_rnn_counters', ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value


def rnn_waited_launches() -> int:
    """persistent launches whose stream first had to wait (on the device) for another stream's persistent launches"""
    return int(_lib.query('ft_rnn_waited_launches'))


def gru_fwd(xp, whh_f, whh_r, bhh_f, bhh_r, H: int, save_gates: bool):
    _chk(xp, 'xp')
    T, B, _ = xp.shape
    out = torch.empty(T, B, 2 * H, device=xp.device, dtype=xp.dtype)
    gates = torch.empty(T, B, 2, 4 * H, device=xp.device, dtype=xp.dtype) if save_gates else None
    ws, nb = _rnn_workspace(3, B, H, xp.device)
    _lib.call('ft_gru_fwd', _p(xp), _p(whh_f), _p(whh_r), _p(bhh_f), _p(bhh_r), _p(out), _p(gates), B, T, H,
              _p(ws), nb, _stream())
    return out, gates


def gru_bwd(dout, out, gates, whhT_f, whhT_r, H: int):
    _chk(dout, 'dout')
    T, B, _ = out.shape
    dxp = torch.empty(T, B, 6 * H, device=out.device, dtype=out.dtype)
    dhp = torch.empty(T, B, 6 * H, device=out.device, dtype=out.dtype)
    carry = torch.empty(B, 2, H, device=out.device, dtype=out.dtype)
    ws, nb = _rnn_workspace(3, B, H, out.device)
    _lib.call('ft_gru_bwd', _p(dout), _p(out), _p(gates), _p(whhT_f), _p(whhT_r), _p(dxp), _p(dhp), _p(carry), B, T,
              H, _p(ws), nb, _stream())
    return dxp, dhp


def lstm_fwd(xp, whh_f, whh_r, bhh_f, bhh_r, lens: Optional[torch.Tensor], H: int, save_gates: bool):
    _chk(xp, 'xp')
    T, B, _ = xp.shape
    raw = torch.empty(T, B, 2 * H, device=xp.device, dtype=xp.dtype)
    cst = torch.empty(T, B, 2 * H, device=xp.device, dtype=xp.dtype)
    gates = torch.empty(T, B, 2, 4 * H, device=xp.device, dtype=xp.dtype) if save_gates else None
    ws, nb = _rnn_workspace(4, B, H, xp.device)
    _lib.call('ft_lstm_fwd', _p(xp), _p(whh_f), _p(whh_r), _p(bhh_f), _p(bhh_r), _p(lens), _p(raw), _p(cst),
              _p(gates), B, T, H, _p(ws), nb, _stream())
    return raw, cst, gates


# ---- recurrent layer forward with the input projection overlapped with the recurrence (ft_*_layer_fwd) ----
_proj_streams = {}


def rnn_overlap(rows: int, T: int) -> int:
    """number of time chunks the input projection of a recurrent layer is cut into (0: projection in front, whole).
    Off by default: measured neutral on the train step (the only layer it applies to, the postnet GRU, has a 130 us
    projection; the 512-wide LSTM fills its XCDs and cannot be overlapped from another stream at all,
    profiles/r03_xcd_dispatch_probe.txt).  FT_RNN_OVERLAP=1 turns it on, FT_RNN_CHUNKS sets the count (default 4)."""
    if os.environ.get('FT_RNN_OVERLAP', '0') != '1' or rows < 8192 or T < 256:
        return 0
    return max(2, int(os.environ.get('FT_RNN_CHUNKS', '4')))


def _proj_stream(device) -> 'torch.cuda.Stream':
    key = torch.device(device).index or 0
    st = _proj_streams.get(key)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _proj_streams[key] = st
    return st


def lstm_layer_fwd(x, wih_f, wih_r, bih_f, bih_r, whh_f, whh_r, bhh_f, bhh_r, lens: Optional[torch.Tensor], H: int,
                   save_gates: bool, nchunks: int):
    """x [B,T,I] batch-major -> (raw, cst, gates) as lstm_fwd; the projection runs in time chunks beside the recurrence"""
    _chk(x, 'x')
    B, T, I = x.shape
    dev = x.device
    xp = torch.empty(T, B, 8 * H, device=dev, dtype=x.dtype)
    raw = torch.empty(T, B, 2 * H, device=dev, dtype=x.dtype)
    cst = torch.empty(T, B, 2 * H, device=dev, dtype=x.dtype)
    gates = torch.empty(T, B, 2, 4 * H, device=dev, dtype=x.dtype) if save_gates else None
    gate = torch.empty(16, device=dev, dtype=torch.int32)
    ws, nb = _rnn_workspace(4, B, H, dev)
    lead = int(os.environ.get('FT_RNN_REV_LEAD', str(nchunks // 2))) if lens is not None else 0
    _lib.call('ft_lstm_layer_fwd', _p(x), I, _p(wih_f), _p(wih_r), _p(bih_f), _p(bih_r), _p(xp), _p(whh_f), _p(whh_r),
              _p(bhh_f), _p(bhh_r), _p(lens), _p(raw), _p(cst), _p(gates), B, T, H, _p(ws), nb, _p(gate), nchunks, lead,
              _stream(), _proj_stream(dev).cuda_stream)
    return raw, cst, gates


def gru_layer_fwd(x, wih_f, wih_r, bih_f, bih_r, whh_f, whh_r, bhh_f, bhh_r, H: int, save_gates: bool, nchunks: int):
    _chk(x, 'x')
    B, T, I = x.shape
    dev = x.device
    xp = torch.empty(T, B, 6 * H, device=dev, dtype=x.dtype)
    out = torch.empty(T, B, 2 * H, device=dev, dtype=x.dtype)
    gates = torch.empty(T, B, 2, 4 * H, device=dev, dtype=x.dtype) if save_gates else None
    gate = torch.empty(16, device=dev, dtype=torch.int32)
    ws, nb = _rnn_workspace(3, B, H, dev)
    _lib.call('ft_gru_layer_fwd', _p(x), I, _p(wih_f), _p(wih_r), _p(bih_f), _p(bih_r), _p(xp), _p(whh_f), _p(whh_r),
              _p(bhh_f), _p(bhh_r), _p(out), _p(gates), B, T, H, _p(ws), nb, _p(gate), nchunks, _stream(),
              _proj_stream(dev).cuda_stream)
    return out, gates


def lstm_bwd(dout, raw, cst, gates, whhT_f, whhT_r, lens: Optional[torch.Tensor], H: int):
    _chk(dout, 'dout')
    T, B, _ = raw.shape
    dg = torch.empty(T, B, 8 * H, device=raw.device, dtype=raw.dtype)
    carry = torch.empty(B, 2, H, device=raw.device, dtype=raw.dtype)
    ws, nb = _rnn_workspace(4, B, H, raw.device)
    _lib.call('ft_lstm_bwd', _p(dout), _p(raw), _p(cst), _p(gates), _p(whhT_f), _p(whhT_r), _p(lens), _p(dg),
              _p(carry), B, T, H, _p(ws), nb, _stream())
    return dg


def fill_padded(raw_tm: torch.Tensor, lens: Optional[torch.Tensor], pad: float) -> torch.Tensor:
    """time-major raw [T,B,C] -> batch-major [B,T,C] with `pad` at t >= lens[b] (lens None: pure layout change)"""
    T, B, C = raw_tm.shape
    out = torch.empty(B, T, C, device=raw_tm.device, dtype=raw_tm.dtype)
    _lib.call('ft_fill_padded', _p(raw_tm), _p(lens), _p(out), B, T, C, pad, _stream())
    return out


def bt_transpose(src: torch.Tensor, to_time_major: bool) -> torch.Tensor:
    """[B,T,C] -> [T,B,C] (to_time_major) or [T,B,C] -> [B,T,C]"""
    _chk(src, 'src')
    if to_time_major:
        B, T, C = src.shape
        dst = torch.empty(T, B, C, device=src.device, dtype=src.dtype)
    else:
        T, B, C = src.shape
        dst = torch.empty(B, T, C, device=src.device, dtype=src.dtype)
    _lib.call('ft_bt_transpose', _p(src), _p(dst), B, T, C, int(to_time_major), _stream())
    return dst


def mask_rows(src: torch.Tensor, lens: torch.Tensor) -> torch.Tensor:
    B, T, C = src.shape
    dst = torch.empty_like(src)
    _lib.call('ft_mask_rows', _p(src), _p(lens), _p(dst), B, T, C, _stream())
    return dst
