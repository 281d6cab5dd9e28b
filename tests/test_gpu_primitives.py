"""GPU parity: GEMM-shaped ops + LengthRegulator through the C ABI vs the CPU oracle."""
import numpy as np
import pytest
import torch

from helpers import load_npz, maxdiff

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def H():
    from forwardtacotron_amd import hip
    assert torch.cuda.is_available()
    return hip


def dev(t):
    return t.cuda().contiguous()


def rel_err(a, b):
    a = torch.as_tensor(a).double().cpu(); b = torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize('rows,in_f,out_f', [(7, 5, 3), (64, 32, 64), (130, 257, 66), (4096, 256, 512),
                                             (1, 1, 1), (300, 1024, 80), (333, 10, 7)])
def test_linear_fwd_bwd(H, rows, in_f, out_f):
    g = torch.Generator().manual_seed(rows + in_f)
    x = torch.randn(rows, in_f, generator=g); w = torch.randn(out_f, in_f, generator=g)
    b = torch.randn(out_f, generator=g); dy = torch.randn(rows, out_f, generator=g)
    y = H.linear_fwd(dev(x), dev(w), dev(b))
    ref = x.double() @ w.double().t() + b.double()
    assert rel_err(y, ref) < 2e-6
    yr = H.linear_fwd(dev(x), dev(w), dev(b), relu=True)
    assert rel_err(yr, ref.clamp_min(0)) < 2e-6
    dx = H.linear_bwd_data(dev(dy), dev(w))
    assert rel_err(dx, dy.double() @ w.double()) < 2e-6
    dw = H.linear_bwd_weight(dev(dy), dev(x))
    assert rel_err(dw, dy.double().t() @ x.double()) < 2e-6


def test_large_gemm_split_path_is_fp32_accurate(H):
    """Shapes big enough for the 128x128 tile, i.e. the bf16-split MFMA kernel (6 bf16 products per fp32 product of an
    exact 3-way operand split): full-mantissa operands over a wide dynamic range, checked against float64 to the
    same relative bar as the f32-MFMA path, forward, data gradient (NT form through the transposed weight) and a
    5-tap convolution with row shifts."""
    g = torch.Generator().manual_seed(77)
    rows, in_f, out_f = 8192, 512, 768
    x = torch.randn(rows, in_f, generator=g) * torch.exp(3 * torch.randn(rows, 1, generator=g))      # rows span ~e^±9
    w = torch.randn(out_f, in_f, generator=g) * torch.exp(torch.randn(out_f, 1, generator=g))
    b = torch.randn(out_f, generator=g)
    y = H.linear_fwd(dev(x), dev(w), dev(b))
    ref = x.double() @ w.double().t() + b.double()
    row_scale = (x.double().abs() @ w.double().abs().t()) + b.double().abs()     # sum |a||b| + |bias|: the error scale
    # f32-class arithmetic lands at ~2e-7 here; a 2-way (hi+lo) split would sit near 3e-6
    assert float(((y.cpu().double() - ref).abs() / row_scale).max()) < 1e-6
    dy = torch.randn(rows, out_f, generator=g)
    dx = H.linear_bwd_data(dev(dy), dev(w))
    refd = dy.double() @ w.double()
    scale_d = dy.double().abs() @ w.double().abs() + 1e-30
    assert float(((dx.cpu().double() - refd).abs() / scale_d).max()) < 1e-6
    B, T, Cin, Cout, k = 32, 600, 128, 256, 5
    xc = torch.randn(B, T, Cin, generator=g)
    wc = torch.randn(Cout, Cin, k, generator=g)
    yc = H.conv1d_fwd(dev(xc), H.conv_pack_weight(dev(wc)), relu=False)
    refc = _conv_ref(xc, wc, False)[:, :T]
    assert rel_err(yc, refc) < 2e-6
    # weight gradients (TN form: the contraction runs over the rows of both operands; r-pair packed LDS tiles)
    dyw = torch.randn(4096, 1024, generator=g) * torch.exp(2 * torch.randn(4096, 1, generator=g))
    xw = torch.randn(4096, 1024, generator=g) * torch.exp(2 * torch.randn(4096, 1, generator=g))
    dw = H.linear_bwd_weight(dev(dyw), dev(xw))
    refw = dyw.double().t() @ xw.double()
    scale_w = dyw.double().abs().t() @ xw.double().abs() + 1e-30
    assert float(((dw.cpu().double() - refw).abs() / scale_w).max()) < 1e-6
    Cin2 = 256
    xc2 = torch.randn(B, T, Cin2, generator=g)
    dyc = torch.randn(B, T, Cout, generator=g)
    dwc = torch.empty(Cout, Cin2, k, device='cuda')
    dyd = dev(dyc)
    H.conv1d_bwd_weight_raw(dyd.data_ptr(), Cout, dev(xc2), dwc, T, T)
    xp = torch.nn.functional.pad(xc2.double(), (0, 0, k // 2, k // 2))
    refdw = torch.stack([torch.einsum('bto,bti->oi', dyc.double(), xp[:, j:j + T]) for j in range(k)], dim=-1)
    assert rel_err(dwc, refdw) < 2e-6


def test_pack_weights_one_launch(H):
    """ft_pack_weights: every pack / transpose in one launch == the per-layer packs, bit for bit (pure data movement);
    ragged dims exercise the tile edges; with the cache installed the wrappers hand out the cached buffers."""
    g = torch.Generator().manual_seed(11)
    mats = [dev(torch.randn(r, c, generator=g)) for r, c in ((64, 32), (33, 65), (1, 7), (384, 80), (100, 257))]
    convs = [dev(torch.randn(co, ci, k, generator=g)) for co, ci, k in ((40, 33, 5), (32, 64, 1), (7, 3, 16))]
    banks = [[dev(torch.randn(24, 17, k, generator=g)) for k in range(1, 6)],
             [dev(torch.randn(8, 8, k, generator=g)) for k in range(1, 3)]]
    cache = H.PackCache(mats, convs, banks, mats[0].device)
    cache.refresh()
    torch.cuda.synchronize()
    for w in mats:
        assert torch.equal(cache.t2d[(w.data_ptr(), *w.shape)], w.t().contiguous())
    for w in convs:
        assert torch.equal(cache.wp[w.data_ptr()], w.permute(2, 0, 1).contiguous())
        assert torch.equal(cache.wpt[w.data_ptr()], w.permute(2, 1, 0).contiguous())
    for ws in banks:
        wp_all, wpt_all = cache.bank[tuple(w.data_ptr() for w in ws)]
        assert torch.equal(wp_all, torch.cat([w.permute(2, 0, 1).reshape(-1) for w in ws]))
        assert torch.equal(wpt_all, torch.cat([w.permute(2, 1, 0).reshape(-1) for w in ws]))
        assert torch.equal(H.bank_packs(ws, False), wp_all) and torch.equal(H.bank_packs(ws, True), wpt_all)
    try:
        H.pack_cache = cache
        assert H.transpose2d(mats[1]).data_ptr() == cache.t2d[(mats[1].data_ptr(), 33, 65)].data_ptr()
        assert H.conv_pack_weight(convs[0]).data_ptr() == cache.wp[convs[0].data_ptr()].data_ptr()
        assert H.conv_pack_weight_t(convs[0]).data_ptr() == cache.wpt[convs[0].data_ptr()].data_ptr()
        assert H.bank_packs(banks[0], True).data_ptr() == cache.bank[tuple(w.data_ptr() for w in banks[0])][1].data_ptr()
        other = dev(torch.randn(5, 6, generator=g))
        assert torch.equal(H.transpose2d(other), other.t().contiguous())      # miss -> packed on the fly
    finally:
        H.pack_cache = None
        cache.release()


def test_linear_multi(H):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 50, 24, generator=g)
    ws = [torch.randn(n, 24, generator=g) for n in (24, 24, 9)]
    bs = [torch.randn(24, generator=g), None, torch.randn(9, generator=g)]
    y = H.linear_multi_fwd(dev(x), [dev(w) for w in ws], [dev(b) if b is not None else None for b in bs])
    ref = torch.cat([x @ w.t() + (b if b is not None else 0) for w, b in zip(ws, bs)], dim=-1)
    assert rel_err(y, ref) < 2e-6


def test_linear_bwd_weight_shift(H):
    from forwardtacotron_amd.hip import linear_bwd_weight_raw
    g = torch.Generator().manual_seed(4)
    B, T, I, O = 3, 11, 6, 9
    dy = torch.randn(B, T, O, generator=g); x = torch.randn(B, T, I, generator=g)
    for shift in (-1, 1):
        xs = torch.zeros_like(x)
        if shift == -1:
            xs[:, 1:] = x[:, :-1]
        else:
            xs[:, :-1] = x[:, 1:]
        ref = dy.reshape(-1, O).double().t() @ xs.reshape(-1, I).double()
        dyd, xd = dev(dy), dev(x)
        dw = torch.empty(O, I, device='cuda')
        linear_bwd_weight_raw(dyd.data_ptr(), O, xd.data_ptr(), I, dw, B * T, I, O, B=B, T=T, x_shift=shift)
        assert rel_err(dw, ref) < 2e-6


def _conv_ref(x_cl, w, relu):
    from oracle import ft_oracle as O
    y = O.conv1d(x_cl.transpose(1, 2).double(), w.double())
    if relu:
        y = y.clamp_min(0)
    return y.transpose(1, 2)     # [B,Tout_full,Cout]


@pytest.mark.parametrize('B,T,Cin,Cout,k', [(2, 9, 6, 8, 5), (3, 11, 5, 7, 4), (2, 17, 16, 16, 1), (1, 5, 3, 4, 2),
                                            (2, 40, 33, 70, 16), (4, 128, 64, 256, 5), (3, 7, 10, 9, 3),
                                            (2, 3, 4, 4, 7)])
def test_conv1d_fwd_bwd(H, B, T, Cin, Cout, k):
    g = torch.Generator().manual_seed(B * 100 + k)
    x = torch.randn(B, T, Cin, generator=g); w = torch.randn(Cout, Cin, k, generator=g)
    wp = H.conv_pack_weight(dev(w))
    assert maxdiff(wp.cpu(), w.permute(2, 0, 1).contiguous()) == 0.0
    full = _conv_ref(x, w, True)
    Tfull = full.shape[1]
    for Tout in sorted({T, Tfull}):
        y = H.conv1d_fwd(dev(x), wp, relu=True, Tout=Tout)
        assert rel_err(y, full[:, :Tout]) < 3e-6, (Tout,)
    sc = torch.rand(Cout, generator=g) + 0.5; sh = torch.randn(Cout, generator=g)
    y = H.conv1d_fwd(dev(x), wp, relu=True, Tout=T, scale=dev(sc), shift=dev(sh))
    assert rel_err(y, full[:, :T] * sc.double() + sh.double()) < 3e-6
    # backward vs autograd of the oracle conv (no relu), using all Tfull rows as valid
    xg = x.double().transpose(1, 2).clone().requires_grad_(True)
    wg = w.double().clone().requires_grad_(True)
    from oracle import ft_oracle as O
    yo = O.conv1d(xg, wg)
    dy = torch.randn(B, Tfull, Cout, generator=g)
    (yo * dy.double().transpose(1, 2)).sum().backward()
    dyd = dev(dy)
    dx = torch.empty(B, T, Cin, device='cuda')
    H.conv1d_bwd_data_raw(dyd.data_ptr(), Cout, wp, dx, B, T, Tfull, Tfull, False)
    assert rel_err(dx, xg.grad.transpose(1, 2)) < 3e-6
    dw = torch.empty(Cout, Cin, k, device='cuda')
    H.conv1d_bwd_weight_raw(dyd.data_ptr(), Cout, dev(x), dw, Tfull, Tfull)
    assert rel_err(dw, wg.grad) < 3e-6
    if Tfull == T + 1:
        # only the first T rows valid (sliced even-k conv)
        xg.grad = None; wg.grad = None
        yo = O.conv1d(xg, wg)[:, :, :T]
        (yo * dy[:, :T].double().transpose(1, 2)).sum().backward()
        H.conv1d_bwd_data_raw(dyd.data_ptr(), Cout, wp, dx, B, T, Tfull, T, False)
        assert rel_err(dx, xg.grad.transpose(1, 2)) < 3e-6
        H.conv1d_bwd_weight_raw(dyd.data_ptr(), Cout, dev(x), dw, Tfull, T)
        assert rel_err(dw, wg.grad) < 3e-6


@pytest.mark.parametrize('B,T,Cin,C,K', [(2, 9, 6, 8, 4), (3, 20, 10, 9, 5), (2, 64, 80, 64, 8), (1, 33, 32, 32, 16)])
def test_conv_bank_fwd(H, B, T, Cin, C, K):
    g = torch.Generator().manual_seed(K)
    x = torch.randn(B, T, Cin, generator=g)
    ws = [torch.randn(C, Cin, k, generator=g) for k in range(1, K + 1)]
    wp_all = torch.cat([H.conv_pack_weight(dev(w)).reshape(-1) for w in ws])
    y = H.conv_bank_fwd(dev(x), wp_all, K, C, relu=True, Tout=T + 1)
    for i, w in enumerate(ws):
        full = _conv_ref(x, w, True)
        n = full.shape[1]
        assert rel_err(y[:, :n, i * C:(i + 1) * C], full) < 3e-6, i


@pytest.mark.parametrize('B,T,Cin,C,K', [(2, 9, 8, 8, 4), (3, 20, 12, 12, 5), (32, 128, 256, 32, 6), (4, 700, 80, 64, 8)])
def test_conv_bank_bwd_data(H, B, T, Cin, C, K):
    """dx of the whole bank vs autograd through the reference formulation (conv -> [:T+1 or :T] rows as the bank
    buffer holds them).  (32,128,256,..) is prenet-shaped: few output tiles -> per-member partials + ordered sum;
    the others take the chained launch; both through the transposed packs (NT form) and the plain ones."""
    g = torch.Generator().manual_seed(100 + K)
    ws = [torch.randn(C, Cin, k, generator=g) for k in range(1, K + 1)]
    dy = torch.randn(B, T + 1, K * C, generator=g)
    x = torch.randn(B, T, Cin, generator=g, dtype=torch.float64, requires_grad=True)
    tot = 0
    for i, w in enumerate(ws):
        k = i + 1
        full = torch.nn.functional.conv1d(x.transpose(1, 2), w.double(), padding=k // 2).transpose(1, 2)
        n = full.shape[1]                     # T (odd k) or T+1 (even k)
        tot = tot + (full * dy[:, :n, i * C:(i + 1) * C].double()).sum()
    tot.backward()
    wd = [dev(w) for w in ws]
    wp_all = torch.cat([H.conv_pack_weight(w).reshape(-1) for w in wd])
    dyd = dev(dy)
    assert rel_err(H.conv_bank_bwd_data(dyd, wp_all, K, C, Cin, T, ws=wd), x.grad) < 3e-6
    assert rel_err(H.conv_bank_bwd_data(dyd, wp_all, K, C, Cin, T), x.grad) < 3e-6


@pytest.mark.parametrize('B,T,Cin,C,K', [(2, 40, 16, 128, 3), (8, 128, 256, 256, 5), (4, 333, 80, 128, 8)])
def test_conv_bank_bwd_weight_one_launch(H, B, T, Cin, C, K):
    """All members' weight gradients from one launch (conv-bank mode of the TN GEMM) vs float64 autograd through
    the reference formulation; also equal (to rounding) to the per-member launches."""
    g = torch.Generator().manual_seed(200 + K)
    x = torch.randn(B, T, Cin, generator=g)
    dy = torch.randn(B, T + 1, K * C, generator=g)
    ws = [torch.zeros(C, Cin, k, dtype=torch.float64, requires_grad=True) for k in range(1, K + 1)]
    tot = 0
    for i, w in enumerate(ws):
        k = i + 1
        full = torch.nn.functional.conv1d(x.double().transpose(1, 2), w, padding=k // 2).transpose(1, 2)
        tot = tot + (full * dy[:, :full.shape[1], i * C:(i + 1) * C].double()).sum()
    tot.backward()
    xd, dyd = dev(x), dev(dy)
    dws = [torch.full((C, Cin, k), float('nan'), device='cuda') for k in range(1, K + 1)]
    H.conv_bank_bwd_weight(dyd, xd, dws, C)
    for i, w in enumerate(ws):
        k = i + 1
        assert rel_err(dws[i], w.grad) < 3e-6, i
        one = torch.empty(C, Cin, k, device='cuda')
        H.conv1d_bwd_weight_raw(dyd.data_ptr() + i * C * 4, K * C, xd, one, T + 1, T + (1 if k % 2 == 0 else 0))
        assert rel_err(dws[i], one) < 2e-6, i


def test_length_regulator_golden(H):
    L = load_npz('layers.npz')
    x = torch.from_numpy(L['lr/x']); dur = torch.from_numpy(L['lr/dur_in'].copy())
    durd = dev(dur)
    cum, total = H.lr_scan(durd)
    assert np.array_equal(durd.cpu().numpy(), L['lr/dur_after'])
    Tm = int(total.max())
    y = H.lr_expand(dev(x), cum, Tm)
    assert np.array_equal(y.cpu().numpy(), L['lr/y'])      # bit-exact


@pytest.mark.parametrize('B,Tx,C,maxd', [(3, 7, 5, 4), (32, 128, 512, 12), (5, 200, 16, 3), (2, 65, 4, 40), (1, 1, 1, 2)])
def test_length_regulator_vs_oracle(H, B, Tx, C, maxd):
    from oracle import ft_oracle as O
    g = torch.Generator().manual_seed(B + Tx)
    x = torch.randn(B, Tx, C, generator=g)
    dur = torch.randint(-1, maxd, (B, Tx), generator=g).float() + torch.rand(B, Tx, generator=g) * 0.99
    xo = x.clone().requires_grad_(True)
    duro = dur.clone()
    yo = O.length_regulate(xo, duro)
    durd = dev(dur)
    cum, total = H.lr_scan(durd)
    assert np.array_equal(durd.cpu().numpy(), duro.numpy())
    src_o, tot_o = O.lr_index_map(duro.numpy())
    assert np.array_equal(total.cpu().numpy(), tot_o.astype(np.int32))
    Tm = int(total.max())
    y, src = H.lr_expand(dev(x), cum, Tm, want_src=True)
    assert np.array_equal(y.cpu().numpy(), yo.detach().numpy())
    assert np.array_equal(src.cpu().numpy(), src_o.astype(np.int32))
    dy = torch.randn(B, Tm, C, generator=g)
    (yo * dy).sum().backward()
    dx = H.lr_bwd(dev(dy), cum, Tx)
    assert maxdiff(dx.cpu(), xo.grad) < 1e-5


def test_length_regulator_all_zero(H):
    durd = torch.zeros(2, 3, device='cuda')
    cum, total = H.lr_scan(durd)
    assert int(total.max()) == 0
    y = H.lr_expand(torch.randn(2, 3, 4, device='cuda'), cum, 0)
    assert tuple(y.shape) == (2, 0, 4)


def test_linear_time_major_layouts(H):
    """GEMMs read / write the recurrences' time-major [T,B,*] buffers without a transposition pass."""
    from forwardtacotron_amd.hip import linear_bwd_weight_raw
    g = torch.Generator().manual_seed(9)
    B, T, I, O = 5, 13, 12, 7
    x = torch.randn(B, T, I, generator=g); w = torch.randn(O, I, generator=g); b = torch.randn(O, generator=g)
    ref = x.double() @ w.double().t() + b.double()                      # [B,T,O]
    x_tm = x.transpose(0, 1).contiguous()
    assert rel_err(H.linear_fwd(dev(x), dev(w), dev(b), y_tm_B=B), ref.transpose(0, 1)) < 2e-6
    assert rel_err(H.linear_fwd(dev(x_tm), dev(w), dev(b), x_tm_B=B), ref) < 2e-6
    assert rel_err(H.linear_fwd(dev(x_tm), dev(w), dev(b), x_tm_B=B, y_tm_B=B), ref.transpose(0, 1)) < 2e-6
    assert rel_err(H.linear_multi_fwd(dev(x), [dev(w), dev(w)], [dev(b), None], y_tm_B=B)[..., :O],
                   ref.transpose(0, 1)) < 2e-6
    dy = torch.randn(B, T, O, generator=g)
    dref = dy.double() @ w.double()
    dy_tm = dy.transpose(0, 1).contiguous()
    assert rel_err(H.linear_bwd_data(dev(dy_tm), dev(w), dy_tm_B=B), dref) < 2e-6
    assert rel_err(H.linear_bwd_data(dev(dy), dev(w), dx_tm_B=B), dref.transpose(0, 1)) < 2e-6
    assert tuple(H.bt_transpose(dev(x), True).shape) == (T, B, I)
    assert torch.equal(H.bt_transpose(H.bt_transpose(dev(x), True), False).cpu(), x)
    for dy_tm_f, x_tm_f, shift in [(True, False, 0), (False, True, 0), (True, True, -1), (True, True, 1)]:
        xs = torch.zeros_like(x)
        if shift == -1:
            xs[:, 1:] = x[:, :-1]
        elif shift == 1:
            xs[:, :-1] = x[:, 1:]
        else:
            xs = x
        wref = dy.reshape(-1, O).double().t() @ xs.reshape(-1, I).double()
        dyd = dev(dy_tm if dy_tm_f else dy); xd = dev(x_tm if x_tm_f else x)
        dw = torch.empty(O, I, device='cuda')
        linear_bwd_weight_raw(dyd.data_ptr(), O, xd.data_ptr(), I, dw, B * T, I, O, B=B, T=T, x_shift=shift,
                              dy_tm=dy_tm_f, x_tm=x_tm_f)
        assert rel_err(dw, wref) < 2e-6, (dy_tm_f, x_tm_f, shift)


@pytest.mark.parametrize('B,T,Cin,C,K', [(2, 9, 8, 8, 4), (3, 37, 12, 12, 5), (4, 700, 80, 64, 8), (32, 128, 256, 128, 3)])
def test_bank_bn_pool_fused_equals_three_passes(H, B, T, Cin, C, K):
    """BatchNorm apply + MaxPool1d(2,1,1) in one pass (ft_bn_pool_from_partials / ft_bn_pool_bwd) against the separate
    kernels AND against torch (common_layers.py:100-105) -- the bank's ReLU leaves runs of equal zeros, so the pooling
    ties (first maximal element takes the gradient) are exercised on every channel."""
    g = torch.Generator().manual_seed(T + C)
    x = torch.randn(B, T, Cin, generator=g)
    ws = [torch.randn(C, Cin, k, generator=g) * 0.2 for k in range(1, K + 1)]
    gamma = torch.randn(K * C, generator=g)          # negative gammas flip the order of z relative to y
    beta = torch.randn(K * C, generator=g)
    dout = torch.randn(B, T, K * C, generator=g)
    wp_all = torch.cat([H.conv_pack_weight(dev(w)).reshape(-1) for w in ws])
    res = {}
    for fused in (True, False):
        rm, rv = torch.zeros(K * C).cuda(), torch.ones(K * C).cuda()
        ybank, part, nch = H.conv_bank_fwd_stats(dev(x), wp_all, K, C, relu=True)
        if fused:
            assert H.bn_pool_fusable(ybank, C)
            out, mean, rstd = H.bn_pool_from_partials(part, nch, ybank, dev(gamma), dev(beta), rm, rv, Tout=T, group=C)
            dy, dg, db = H.bn_pool_bwd(dev(dout), ybank, dev(gamma), dev(beta), mean, rstd, group=C, relu=True)
        else:
            z, mean, rstd = H.bn_train_from_partials(part, nch, ybank, dev(gamma), dev(beta), rm, rv, Tout=T, group=C)
            out = H.maxpool2_fwd(z)
            dy, dg, db = H.bn_bwd(H.maxpool2_bwd(dev(dout), z), ybank, dev(gamma), mean, rstd, group=C, relu=True)
        res[fused] = [t.cpu() for t in (out, dy, dg, db, rm, rv)]
    for a, b, name in zip(res[True], res[False], ('out', 'dy', 'dgamma', 'dbeta', 'running_mean', 'running_var')):
        assert torch.equal(a, b), name                # same arithmetic, same summation order: bit-equal
    # torch on the CPU, float64
    xs = x.double().transpose(1, 2)
    outs = []
    for i, w in enumerate(ws):
        k = i + 1
        y = torch.relu(torch.nn.functional.conv1d(xs, w.double(), padding=k // 2))
        y = torch.nn.functional.batch_norm(y, None, None, gamma[i * C:(i + 1) * C].double(), beta[i * C:(i + 1) * C].double(),
                                           training=True, eps=1e-5)
        outs.append(y[:, :, :T])
    zc = torch.cat(outs, dim=1)
    ref = torch.nn.functional.max_pool1d(zc, kernel_size=2, stride=1, padding=1)[:, :, :T]
    assert rel_err(res[True][0], ref.transpose(1, 2).detach()) < 2e-5
    # (the gradients' parity with the reference is pinned end to end: tests/test_gpu_model.py's golden train steps)




def test_non_recurrent_entry_points_are_graph_capturable(H):
    """include/fwdtaco_hip.h, conventions: an entry point only enqueues work on the stream it is given (no sync, no
    allocation), so a chain of them can be captured in a hipGraph -- everything except the persistent recurrences, whose
    admission bookkeeping records and queries events on the host.  A token-side chain (embedding -> k-tap conv + ReLU ->
    Linear -> LengthRegulator scan + expand) is captured once with torch.cuda.graph and replayed on new inputs: the
    replays must equal the eager launches bit for bit."""
    g = torch.Generator().manual_seed(9)
    V, C, B, Tx, Tm = 40, 64, 4, 24, 100
    emb = dev(torch.randn(V, C, generator=g))
    cw = dev(torch.randn(96, C, 5, generator=g) * 0.2)
    lw = dev(torch.randn(48, 96, generator=g) * 0.2)
    wp = H.conv_pack_weight(cw)
    idx = dev(torch.randint(1, V, (B, Tx), generator=g))
    dur = dev(torch.randint(0, 6, (B, Tx), generator=g).float())

    def chain(idx_, dur_):
        x = H.embedding_fwd(idx_, emb)
        y = H.linear_fwd(H.conv1d_fwd(x, wp, True), lw)
        cum, total = H.lr_scan(dur_)
        return H.lr_expand(y, cum, Tm), total

    s_idx, s_dur = idx.clone(), dur.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        chain(s_idx, s_dur)                          # warm-up outside the capture (workspaces, lazy module loads)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out, total = chain(s_idx, s_dur)
    for seed in (1, 2):
        g2 = torch.Generator().manual_seed(seed)
        s_idx.copy_(dev(torch.randint(1, V, (B, Tx), generator=g2)))
        s_dur.copy_(dev(torch.randint(0, 6, (B, Tx), generator=g2).float()))
        graph.replay()
        want, want_total = chain(s_idx.clone(), s_dur.clone())
        torch.cuda.synchronize()
        assert torch.equal(out, want) and torch.equal(total, want_total)


def _tn_pipelined():
    from forwardtacotron_amd import _lib
    return _lib.query('ft_gemm_tn_pipelined_launches')


@pytest.mark.parametrize('form', ['linear_tm_shift', 'conv_k5', 'bank', 'linear_bf16'])
def test_weight_gradient_pipelined_kernel(H, form):
    """ft_gemm_tn_b3p_kernel (the software-pipelined 128x128 weight-gradient kernel) only takes launches with >= 512
    contraction rows per split, which the mid-size model tests do not reach: here every form it serves is run at a size
    that does (the launch counter proves it) and compared with a float64 reference -- the LSTM W_hh form (time-major
    dy, a one-step shift of the time-major x inside each item, T not a multiple of the 16-row stage), a 5-tap conv
    (rows outside the tap's window at both ends of every item), the conv-bank mode (members with their own taps /
    shifts / valid lengths in one launch) and the bf16 precision mode."""
    from forwardtacotron_amd.hip import linear_bwd_weight_raw
    g = torch.Generator().manual_seed(77)
    n0 = _tn_pipelined()
    if form in ('linear_tm_shift', 'linear_bf16'):
        B, T, O, I = 16, 601, 1024, 512
        dy = dev(torch.randn(T, B, O, generator=g))              # time-major, as the recurrences leave it
        x = dev(torch.randn(T, B, I, generator=g))
        old = H.set_gemm_precision('bf16' if form == 'linear_bf16' else 'fp32')
        try:
            for shift in (-1, 1):
                xs = torch.zeros_like(x)
                if shift == -1:
                    xs[1:] = x[:-1]
                else:
                    xs[:-1] = x[1:]
                ref = dy.reshape(-1, O).double().t() @ xs.reshape(-1, I).double()
                dw = torch.empty(O, I, device='cuda')
                linear_bwd_weight_raw(dy.data_ptr(), O, x.data_ptr(), I, dw, B * T, I, O, B=B, T=T, x_shift=shift,
                                      dy_tm=True, x_tm=True)
                assert rel_err(dw, ref) < (2e-2 if form == 'linear_bf16' else 2e-6), shift
        finally:
            H.set_gemm_precision(old)
        assert _tn_pipelined() - n0 == 2
    elif form == 'conv_k5':
        B, T, Cin, Cout, k = 16, 501, 256, 512, 5
        x = dev(torch.randn(B, T, Cin, generator=g))
        dy = dev(torch.randn(B, T, Cout, generator=g))
        dw = torch.empty(Cout, Cin, k, device='cuda')
        H.conv1d_bwd_weight_raw(dy.data_ptr(), Cout, x, dw, T, T)
        for j in range(k):                                        # dw[:, :, j] = sum_{b,t} dy[b,t,:]^T x[b,t+j-k//2,:]
            sh = j - k // 2
            xs = torch.zeros_like(x)
            if sh < 0:
                xs[:, -sh:] = x[:, :sh]
            elif sh > 0:
                xs[:, :-sh] = x[:, sh:]
            else:
                xs = x
            ref = dy.reshape(-1, Cout).double().t() @ xs.reshape(-1, Cin).double()
            assert rel_err(dw[:, :, j], ref) < 2e-6, j
        assert _tn_pipelined() - n0 == 1
    else:
        B, T, Cin, C, K = 64, 2800, 128, 128, 3
        x = dev(torch.randn(B, T, Cin, generator=g))
        dy = dev(torch.randn(B, T + 1, K * C, generator=g))
        dws = [torch.full((C, Cin, kk), float('nan'), device='cuda') for kk in range(1, K + 1)]
        H.conv_bank_bwd_weight(dy, x, dws, C)
        xp = torch.zeros(B, T + 4, Cin, device='cuda')            # x with two zero frames on either side
        xp[:, 2:T + 2] = x
        for i in range(K):
            kk = i + 1
            Tv = T + (1 if kk % 2 == 0 else 0)                    # rows of the bank buffer member kk really produced
            dyk = dy[:, :Tv, i * C:(i + 1) * C].double()
            for j in range(kk):                                   # y[t] = sum_j w[:, :, j] x[t + j - kk//2]
                sh = j - kk // 2
                xs = xp[:, 2 + sh:2 + sh + Tv].double()
                ref = dyk.reshape(-1, C).t() @ xs.reshape(-1, Cin)
                assert rel_err(dws[i][:, :, j], ref) < 3e-6, (kk, j)
        assert _tn_pipelined() - n0 == 1


# ---------------------------------------------------------------------------------------------------
# HighwayNetwork with the gate inside the GEMM epilogues (common_layers.py:35-40; ops.HighwayStackFn)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('rows,C,layers', [(300, 64, 3), (4096, 256, 4), (25000, 128, 2), (77, 32, 1), (26912, 256, 2)])
def test_highway_stack_gates_in_the_gemm_epilogues(H, rows, C, layers, monkeypatch):
    """ops.highway_stack (forward gate = epilogue of the interleaved W1 | W2 product, gate gradient of layer i - 1 =
    epilogue of layer i's data-gradient product) against float64 math and against the unfused per-layer form: rows /
    widths that take the 64-column tiling (gate through LDS between the two waves of a tile row) and the 128-column
    split kernel (both pre-activations in one lane), odd row counts, 1 .. 4 layers."""
    from forwardtacotron_amd import ops
    g = torch.Generator().manual_seed(rows + C)

    class Hw:                                    # parameter containers like model.HighwayNetwork
        def __init__(self):
            self.W1 = torch.nn.Linear(C, C)
            self.W2 = torch.nn.Linear(C, C)
            for p in (self.W1.weight, self.W2.weight):
                p.data = torch.randn(C, C, generator=g) * (1.0 / C ** 0.5)
            for p in (self.W1.bias, self.W2.bias):
                p.data = torch.randn(C, generator=g) * 0.3
            self.W1.cuda(); self.W2.cuda()

    hs = [Hw() for _ in range(layers)]
    x = torch.randn(rows, C, generator=g)
    w = torch.randn(rows, C, generator=g)
    res = {}
    for fused in ('1', '0'):
        monkeypatch.setenv('FT_HIGHWAY_FUSED', fused)
        for h in hs:
            for p in (h.W1.weight, h.W1.bias, h.W2.weight, h.W2.bias):
                p.grad = None
        xg = x.cuda().requires_grad_(True)
        y = ops.highway_stack(xg, hs)
        (y * w.cuda()).sum().backward()
        res[fused] = (y.detach().cpu(), xg.grad.cpu(),
                      [p.grad.cpu() for h in hs for p in (h.W1.weight, h.W1.bias, h.W2.weight, h.W2.bias)])
    xo = x.double().requires_grad_(True)
    Po = [[p.detach().cpu().double().requires_grad_(True) for p in (h.W1.weight, h.W1.bias, h.W2.weight, h.W2.bias)]
          for h in hs]
    yo = xo
    for w1, b1, w2, b2 in Po:
        gt = torch.sigmoid(yo @ w2.t() + b2)
        yo = gt * torch.relu(yo @ w1.t() + b1) + (1. - gt) * yo
    (yo * w.double()).sum().backward()
    # against float64: a ReLU pre-activation within fp32 rounding of zero flips its 0 / 1 derivative (a handful of the
    # rows x C x layers decisions), and one flip changes a whole row of dx -- 99.5 % of the elements must agree
    def close(got, want, tol):
        err = ((got.double() - want.double()).abs() / max(1.0, float(want.abs().max()))).flatten()
        return float(err.kthvalue(max(1, int(err.numel() * 0.995)))[0]) < tol
    for tag in ('1', '0'):
        y, dx, gr = res[tag]
        assert maxdiff(y, yo.detach()) < 2e-5 * max(1.0, float(yo.abs().max())), tag
        assert close(dx, xo.grad, 2e-5), tag
        for got, want in zip(gr, [p.grad for ps in Po for p in ps]):
            assert close(got, want, 1e-4), tag
    # against the per-layer form (pinned by the reference's goldens, test_maxpool_highway_golden / the model tests): the
    # same products in the same order and the same gate arithmetic
    assert maxdiff(res['1'][0], res['0'][0]) <= 1e-6 * max(1.0, float(yo.abs().max()))
    assert maxdiff(res['1'][1], res['0'][1]) <= 1e-6 * max(1.0, float(xo.grad.abs().max()))
    for a, b in zip(res['1'][2], res['0'][2]):
        assert maxdiff(a, b) <= 1e-6 * max(1.0, float(b.abs().max()))


def test_batched_column_sums_are_bit_identical_to_one_call_each(H):
    """ft_colsum_batch: one partial + one finalize launch for up to 16 matrices of equal row count, every sum with the
    chunking and order of its own ft_colsum call (the FFT blocks' eight bias / LayerNorm gradients per block)"""
    g = torch.Generator().manual_seed(5)
    for rows, widths in ((26912, (256, 256, 256, 1024, 256, 256, 256, 768)), (4096, (128, 128, 384)), (37, (64, 4, 260)),
                         (1, (8,))):
        xs = [torch.randn(rows, c, generator=g).cuda() for c in widths]
        got = H.colsum_batch(xs)
        for x, s in zip(xs, got):
            assert torch.equal(s, H.colsum(x))
            assert maxdiff(s.cpu(), x.double().sum(0).cpu()) <= 1e-5 * max(1.0, float(x.abs().sum(0).max()))
    # widths that cannot take the 16-byte path fall back to one call each
    xs = [torch.randn(50, c, generator=g).cuda() for c in (6, 64)]
    for x, s in zip(xs, H.colsum_batch(xs)):
        assert torch.equal(s, H.colsum(x))
