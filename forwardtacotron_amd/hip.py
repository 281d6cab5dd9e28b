"""Thin tensor-level wrappers over the C ABI (include/fwdtaco_hip.h).

torch is used here only as the owner of device memory and of the HIP stream; every computation is a
hand-written gfx950 kernel behind libfwdtaco_hip.so.  All functions require contiguous fp32 CUDA(HIP)
tensors and raise if handed anything else -- there is no fallback path.
"""
import ctypes
from typing import List, Optional, Sequence

import torch

from . import _lib

c_void_p = ctypes.c_void_p


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
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
    """Grow-only scratch buffer per device (safe: every kernel runs on the current stream, in order)."""
    key = (torch.device(device).index or 0)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


def _ptr_array(ts: Sequence[Optional[torch.Tensor]]):
    arr = (ctypes.c_void_p * len(ts))(*[_p(t) for t in ts])
    return arr


# ---------------------------------------------------------------------------------------------------
# linear
# ---------------------------------------------------------------------------------------------------
def linear_fwd(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, relu: bool = False,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [..., in] , w [out, in] -> [..., out]"""
    _chk(x, 'x'); _chk(w, 'w')
    in_f = x.shape[-1]
    rows = x.numel() // max(in_f, 1) if in_f else 0
    out_f = w.shape[0]
    assert w.shape[1] == in_f
    y = out if out is not None else torch.empty(*x.shape[:-1], out_f, device=x.device, dtype=x.dtype)
    _lib.call('ft_linear_fwd', _p(x), in_f, _p(w), _p(bias), _p(y), out_f, rows, in_f, out_f, int(relu), 0,
              _stream())
    return y


def linear_multi_fwd(x: torch.Tensor, ws: List[torch.Tensor], biases: Optional[List[Optional[torch.Tensor]]],
                     relu: bool = False) -> torch.Tensor:
    """Several Linear layers on one input, outputs concatenated along the last dim (one launch)."""
    _chk(x, 'x')
    in_f = x.shape[-1]
    rows = x.numel() // max(in_f, 1)
    outs = [int(w.shape[0]) for w in ws]
    offs = [sum(outs[:i]) for i in range(len(outs))]
    y = torch.empty(*x.shape[:-1], sum(outs), device=x.device, dtype=x.dtype)
    n = len(ws)
    wa = _ptr_array(ws)
    ba = _ptr_array(biases) if biases is not None else None
    oa = (ctypes.c_int * n)(*offs)
    fa = (ctypes.c_int * n)(*outs)
    _lib.call('ft_linear_multi_fwd', _p(x), in_f, n, ctypes.cast(wa, c_void_p),
              ctypes.cast(ba, c_void_p) if ba is not None else None, _p(y), sum(outs),
              ctypes.cast(oa, c_void_p), ctypes.cast(fa, c_void_p), rows, in_f, int(relu), _stream())
    return y


def linear_bwd_data(dy: torch.Tensor, w: torch.Tensor, lddy: Optional[int] = None, out_f: Optional[int] = None,
                    dx: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """dx = dy[..., :out_f] @ w ; dy may be a column slice of a wider buffer (pass lddy/out_f + a pointer view)."""
    _chk(w, 'w')
    out_f = out_f if out_f is not None else dy.shape[-1]
    lddy = lddy if lddy is not None else dy.shape[-1]
    in_f = w.shape[1]
    rows = dy.numel() // dy.shape[-1] if lddy == dy.shape[-1] else None
    assert rows is not None
    if dx is None:
        dx = torch.empty(*dy.shape[:-1], in_f, device=dy.device, dtype=dy.dtype)
    _lib.call('ft_linear_bwd_data', _p(dy), lddy, _p(w), _p(dx), in_f, rows, in_f, out_f, int(accumulate), _stream())
    return dx


def linear_bwd_data_raw(dy_ptr: int, lddy: int, w: torch.Tensor, dx: torch.Tensor, rows: int, out_f: int,
                        accumulate: bool) -> None:
    in_f = w.shape[1]
    _lib.call('ft_linear_bwd_data', dy_ptr, lddy, _p(w), _p(dx), in_f, rows, in_f, out_f, int(accumulate), _stream())


def linear_bwd_weight_raw(dy_ptr: int, lddy: int, x_ptr: int, ldx: int, dw: torch.Tensor, rows: int, in_f: int,
                          out_f: int, B: int = 1, T: int = 0, x_shift: int = 0, accumulate: bool = False) -> None:
    nbytes = _lib.query('ft_linear_bwd_weight_workspace', rows, in_f, out_f)
    ws = workspace(nbytes, dw.device)
    _lib.call('ft_linear_bwd_weight', dy_ptr, lddy, x_ptr, ldx, _p(dw), rows, in_f, out_f, B, T if T else rows,
              x_shift, int(accumulate), _p(ws), ws.numel(), _stream())


def linear_bwd_weight(dy: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """dw[out,in] = dy^T x over all leading positions."""
    _chk(dy, 'dy'); _chk(x, 'x')
    out_f, in_f = dy.shape[-1], x.shape[-1]
    rows = x.numel() // max(in_f, 1)
    dw = torch.empty(out_f, in_f, device=x.device, dtype=x.dtype)
    linear_bwd_weight_raw(_p(dy), out_f, _p(x), in_f, dw, rows, in_f, out_f)
    return dw


# ---------------------------------------------------------------------------------------------------
# channels-last conv
# ---------------------------------------------------------------------------------------------------
def conv_pack_weight(w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[Cout,Cin,k] -> tap-major [k,Cout,Cin]"""
    _chk(w, 'w')
    Cout, Cin, k = w.shape
    wp = out if out is not None else torch.empty(k, Cout, Cin, device=w.device, dtype=w.dtype)
    _lib.call('ft_conv_pack_weight', _p(w), _p(wp), Cout, Cin, k, _stream())
    return wp


def conv1d_fwd(x: torch.Tensor, wp: torch.Tensor, relu: bool, Tout: Optional[int] = None,
               scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [B,T,Cin], wp [k,Cout,Cin] -> y [B,Tout,Cout]"""
    _chk(x, 'x'); _chk(wp, 'wp')
    B, T, Cin = x.shape
    k, Cout, _ = wp.shape
    Tout = T if Tout is None else Tout
    y = torch.empty(B, Tout, Cout, device=x.device, dtype=x.dtype)
    _lib.call('ft_conv1d_fwd', _p(x), Cin, _p(wp), _p(scale), _p(shift), _p(y), Cout, B, T, Cin, Cout, k, Tout,
              int(relu), _stream())
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


def conv1d_bwd_data_raw(dy_ptr: int, lddy: int, wp: torch.Tensor, dx: torch.Tensor, B: int, T: int, Tbuf: int,
                        Tvalid: int, accumulate: bool) -> None:
    k, Cout, Cin = wp.shape
    _lib.call('ft_conv1d_bwd_data', dy_ptr, lddy, _p(wp), _p(dx), Cin, B, T, Cin, Cout, k, Tbuf, Tvalid,
              int(accumulate), _stream())


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
