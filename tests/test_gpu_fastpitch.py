"""GPU parity of the FastPitch variant: attention / LayerNorm / PositionalEncoding building blocks
vs explicit fp32 torch-CPU math, the whole model vs goldens captured from the imported reference
(tests/golden/make_golden_fastpitch.py) and vs the oracle at an awkward size."""
import math

import pytest
import torch

from helpers import TINY_FP, TRAIN_CFG, fp_state, load_npz, sub, maxdiff

pytestmark = pytest.mark.gpu


def _bg(kind, A, lda, sA, Bm, ldb, sB, C, ldc, sC, M, N, K, nb0, nb1):
    from forwardtacotron_amd.fastpitch import _bgemm
    _bgemm(kind, A, lda, sA[0], sA[1], Bm, ldb, sB[0], sB[1], C, ldc, sC[0], sC[1], M, N, K, nb0, nb1,
           torch.device('cuda'))


@pytest.mark.parametrize('B,nh,T,hd', [(2, 2, 9, 4), (3, 1, 37, 8), (2, 2, 130, 64), (1, 4, 64, 7)])
def test_strided_batch_gemms(B, nh, T, hd):
    """QK^T (NT), PV (NN) and P^T dO (TN) straight out of / into the interleaved [B,T,3d] projection buffer.
"""
    g = torch.Generator().manual_seed(B * 100 + T)
    d = nh * hd
    qkv = torch.randn(B, T, 3 * d, generator=g)
    P = torch.randn(B, nh, T, T, generator=g)
    q = qkv[..., :d].reshape(B, T, nh, hd).permute(0, 2, 1, 3)
    k = qkv[..., d:2 * d].reshape(B, T, nh, hd).permute(0, 2, 1, 3)
    v = qkv[..., 2 * d:].reshape(B, T, nh, hd).permute(0, 2, 1, 3)
    dq = qkv.cuda()
    dP = P.cuda()
    S = torch.empty(B, nh, T, T, device='cuda')
    _bg('nt', dq.data_ptr(), 3 * d, (T * 3 * d, hd), dq.data_ptr() + 4 * d, 3 * d, (T * 3 * d, hd), S.data_ptr(), T,
        (nh * T * T, T * T), T, T, hd, B, nh)
    tol = 2e-5 * math.sqrt(max(hd, T))
    assert maxdiff(S.cpu(), q @ k.transpose(-1, -2)) < tol
    att = torch.full((B, T, d), float('nan'), device='cuda')
    _bg('nn', dP.data_ptr(), T, (nh * T * T, T * T), dq.data_ptr() + 8 * d, 3 * d, (T * 3 * d, hd), att.data_ptr(), d,
        (T * d, hd), T, hd, T, B, nh)
    want = (P @ v).permute(0, 2, 1, 3).reshape(B, T, d)
    assert maxdiff(att.cpu(), want) < tol
    out = torch.full((B, T, 3 * d), float('nan'), device='cuda')
    _bg('tn', dP.data_ptr(), T, (nh * T * T, T * T), dq.data_ptr(), 3 * d, (T * 3 * d, hd), out.data_ptr() + 4 * d,
        3 * d, (T * 3 * d, hd), T, hd, T, B, nh)
    want = (P.transpose(-1, -2) @ q).permute(0, 2, 1, 3).reshape(B, T, d)
    assert maxdiff(out[..., d:2 * d].cpu(), want) < tol
    assert torch.isnan(out[..., :d]).all() and torch.isnan(out[..., 2 * d:]).all()      # neighbours untouched


@pytest.mark.parametrize('B,nh,T,hd', [(2, 2, 37, 8), (2, 2, 841, 64), (1, 3, 130, 32), (1, 1, 7, 4), (19, 2, 841, 128)])
def test_row_padded_attention_gemms(B, nh, T, hd):
    """PV (NN) and P^T dO (TN) with the [T,T] operand stored at a row stride rounded up to 4 and flagged as padded
    (the aligned 16-B-load paths; T = 841 is the benchmark's frame count): the pad columns hold NaNs here to prove that
    whatever lies behind column T-1 is ignored (NN) or only reaches masked outputs (TN).  The last case is the flagship
    FastPitch's frame-side attention (head width 128) at a batch that sends the TN product to the software-pipelined
    128x128 kernel in its strided-batch form (the launch counter proves it)."""
    from forwardtacotron_amd import _lib
    from forwardtacotron_amd.fastpitch import _bgemm
    n_pipe0 = _lib.query('ft_gemm_tn_pipelined_launches')
    g = torch.Generator().manual_seed(B * 10 + T)
    d = nh * hd
    Tp = (T + 3) // 4 * 4
    qkv = torch.randn(B, T, 3 * d, generator=g)
    P = torch.randn(B, nh, T, T, generator=g)
    q = qkv[..., :d].reshape(B, T, nh, hd).permute(0, 2, 1, 3)
    v = qkv[..., 2 * d:].reshape(B, T, nh, hd).permute(0, 2, 1, 3)
    Pp = torch.full((B, nh, T, Tp), float('nan'))
    Pp[..., :T] = P
    dPp, dq = Pp.cuda(), qkv.cuda()
    dev = torch.device('cuda')
    tol = 2e-5 * math.sqrt(max(hd, T))
    att = torch.full((B, T, d), float('nan'), device='cuda')
    _bgemm('nn', dPp.data_ptr(), Tp, nh * T * Tp, T * Tp, dq.data_ptr() + 8 * d, 3 * d, T * 3 * d, hd, att.data_ptr(), d,
           T * d, hd, T, hd, T, B, nh, dev, padded=True)
    assert maxdiff(att.cpu(), (P @ v).permute(0, 2, 1, 3).reshape(B, T, d)) < tol
    out = torch.full((B, T, 3 * d), float('nan'), device='cuda')
    _bgemm('tn', dPp.data_ptr(), Tp, nh * T * Tp, T * Tp, dq.data_ptr(), 3 * d, T * 3 * d, hd, out.data_ptr() + 4 * d,
           3 * d, T * 3 * d, hd, T, hd, T, B, nh, dev, padded=True)
    want = (P.transpose(-1, -2) @ q).permute(0, 2, 1, 3).reshape(B, T, d)
    assert maxdiff(out[..., d:2 * d].cpu(), want) < tol
    assert torch.isnan(out[..., :d]).all() and torch.isnan(out[..., 2 * d:]).all()
    if hd == 128:
        assert _lib.query('ft_gemm_tn_pipelined_launches') - n_pipe0 == 1


@pytest.mark.parametrize('B,T,d,nh,masked', [(3, 9, 16, 2, True), (2, 70, 32, 4, True), (2, 33, 24, 3, False)])
def test_attention_matches_torch_mha(B, T, d, nh, masked):
    from forwardtacotron_amd.fastpitch import MHAFn
    torch.manual_seed(B * T)
    mha = torch.nn.MultiheadAttention(d, nh, dropout=0.0)
    with torch.no_grad():
        mha.in_proj_bias.normal_(0, 0.2)
        mha.out_proj.bias.normal_(0, 0.2)
    x = torch.randn(B, T, d)
    w = torch.randn(B, T, d)
    pad = None
    if masked:
        lens = torch.randint(1, T + 1, (B,))
        lens[0] = T
        pad = torch.arange(T).unsqueeze(0) >= lens.unsqueeze(1)
    xr = x.clone().requires_grad_(True)
    y_ref = mha(xr.transpose(0, 1), xr.transpose(0, 1), xr.transpose(0, 1), key_padding_mask=pad)[0].transpose(0, 1)
    (y_ref * w).sum().backward()
    ps = [p.detach().clone().cuda().requires_grad_(True) for p in
          (mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias)]
    xg = x.clone().cuda().requires_grad_(True)
    y = MHAFn.apply(xg, pad.to(torch.uint8).cuda() if masked else None, *ps, nh, 0.0, 0)
    (y * w.cuda()).sum().backward()
    assert maxdiff(y.detach().cpu(), y_ref.detach()) < 2e-5
    assert maxdiff(xg.grad.cpu(), xr.grad) < 5e-5
    for p, r in zip(ps, (mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias)):
        assert maxdiff(p.grad.cpu(), r.grad) < 1e-4


def test_attention_dropout_is_consistent_between_forward_and_backward():
    """With p>0 the backward pass regenerates the same keep-mask: the gradient must equal a finite difference of the
    (seeded, hence deterministic) forward."""
    from forwardtacotron_amd.fastpitch import MHAFn
    torch.manual_seed(3)
    B, T, d, nh = 2, 12, 16, 2
    mha = torch.nn.MultiheadAttention(d, nh)
    ps = [p.detach().clone().cuda() for p in (mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight,
                                               mha.out_proj.bias)]
    x = torch.randn(B, T, d).cuda().requires_grad_(True)
    w = torch.randn(B, T, d).cuda()
    f = lambda t: (MHAFn.apply(t, None, *ps, nh, 0.3, 1234) * w).sum()
    y0 = f(x)
    y0.backward()
    assert float((MHAFn.apply(x, None, *ps, nh, 0.3, 1234) - MHAFn.apply(x, None, *ps, nh, 0.0, 0)).abs().max()) > 1e-3
    dirn = torch.randn_like(x)
    eps = 1e-2
    fd = (float(f(x.detach() + eps * dirn)) - float(f(x.detach() - eps * dirn))) / (2 * eps)
    an = float((x.grad * dirn).sum())
    assert abs(fd - an) < 2e-2 * max(1.0, abs(an))


@pytest.mark.parametrize('rows,D,res', [(27, 16, True), (300, 256, True), (10, 7, False), (65, 130, False)])
def test_add_layernorm(rows, D, res):
    from forwardtacotron_amd.fastpitch import AddLayerNormFn
    g = torch.Generator().manual_seed(rows + D)
    x = torch.randn(3, rows, D, generator=g)
    r = torch.randn(3, rows, D, generator=g) if res else None
    gamma = 1 + 0.2 * torch.randn(D, generator=g)
    beta = 0.1 * torch.randn(D, generator=g)
    w = torch.randn(3, rows, D, generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (x, gamma, beta)] + ([r.clone().requires_grad_(True)] if res else [])
    s = leaves[0] + leaves[3] if res else leaves[0]
    y_ref = torch.nn.functional.layer_norm(s, (D,), leaves[1], leaves[2], 1e-5)
    (y_ref * w).sum().backward()
    cl = [t.clone().cuda().requires_grad_(True) for t in (x, gamma, beta)] + ([r.clone().cuda().requires_grad_(True)] if res else [])
    y = AddLayerNormFn.apply(cl[0], cl[3] if res else None, cl[1], cl[2], 1e-5)
    (y * w.cuda()).sum().backward()
    assert maxdiff(y.detach().cpu(), y_ref.detach()) < 5e-6
    for a, b in zip(cl, leaves):
        assert maxdiff(a.grad.cpu(), b.grad) < 5e-5 * max(1.0, float(b.grad.abs().max()))


@pytest.mark.parametrize('rows,D,p', [(33, 24, 0.3), (300, 384, 0.1)])
def test_add_layernorm_with_fused_dropout_equals_separate_dropout(rows, D, p):
    """norm(x + dropout(res)) in one kernel == DropoutFn (same seed, same counter-based mask) followed by the plain
    AddLayerNorm: outputs and all four gradients identical (pure refactoring of where the mask is applied)."""
    from forwardtacotron_amd.fastpitch import AddLayerNormFn
    from forwardtacotron_amd.ops import DropoutFn
    g = torch.Generator().manual_seed(rows * 3 + D)
    x, r, w = (torch.randn(2, rows, D, generator=g) for _ in range(3))
    gamma = 1 + 0.2 * torch.randn(D, generator=g)
    beta = 0.1 * torch.randn(D, generator=g)
    seed = 987654321
    outs = []
    for fused in (True, False):
        xs, rs, gs, bs = (t.clone().cuda().requires_grad_(True) for t in (x, r, gamma, beta))
        if fused:
            y = AddLayerNormFn.apply(xs, rs, gs, bs, 1e-5, p, seed)
        else:
            y = AddLayerNormFn.apply(xs, DropoutFn.apply(rs, p, seed), gs, bs, 1e-5)
        (y * w.cuda()).sum().backward()
        outs.append([y.detach().cpu()] + [t.grad.cpu() for t in (xs, rs, gs, bs)])
    assert float((outs[0][2] == 0).float().mean()) > 0.5 * p            # the mask really dropped gradient entries
    for a, b in zip(*outs):
        assert maxdiff(a, b) < 1e-6 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize('Cin,Cout,k,relu', [(16, 24, 5, True), (24, 16, 1, False), (7, 9, 3, True), (256, 1024, 9, True)])
def test_conv_bias(Cin, Cout, k, relu):
    from forwardtacotron_amd.fastpitch import ConvBiasFn
    torch.manual_seed(Cin + k)
    B, T = 3, 21
    conv = torch.nn.Conv1d(Cin, Cout, k, padding=k // 2)
    x = torch.randn(B, T, Cin)
    w = torch.randn(B, T, Cout)
    xr = x.clone().requires_grad_(True)
    y_ref = conv(xr.transpose(1, 2))
    if relu:
        y_ref = torch.relu(y_ref)
    y_ref = y_ref.transpose(1, 2)
    (y_ref * w).sum().backward()
    xg = x.clone().cuda().requires_grad_(True)
    cw = conv.weight.detach().clone().cuda().requires_grad_(True)
    cb = conv.bias.detach().clone().cuda().requires_grad_(True)
    y = ConvBiasFn.apply(xg, cw, cb, relu)
    (y * w.cuda()).sum().backward()
    tol = 3e-5 * math.sqrt(Cin * k / 16 + 1)
    assert maxdiff(y.detach().cpu(), y_ref.detach()) < tol
    assert maxdiff(xg.grad.cpu(), xr.grad) < 4 * tol
    assert maxdiff(cw.grad.cpu(), conv.weight.grad) < 4 * tol
    assert maxdiff(cb.grad.cpu(), conv.bias.grad) < 4 * tol


def test_positional_encoding():
    from forwardtacotron_amd.fastpitch import PositionalEncoding
    from helpers import sinusoid_pe
    pe = PositionalEncoding(12, dropout=0.0)
    assert torch.equal(pe.pe, sinusoid_pe(12))
    pe = pe.cuda()
    with torch.no_grad():
        pe.scale.fill_(0.7)
    x = torch.randn(2, 30, 12)
    xg = x.clone().cuda().requires_grad_(True)
    y = pe(xg)
    w = torch.randn(2, 30, 12)
    (y * w.cuda()).sum().backward()
    want = x + 0.7 * sinusoid_pe(12)[:30, 0].unsqueeze(0)
    assert maxdiff(y.detach().cpu(), want) < 1e-6
    assert maxdiff(xg.grad.cpu(), w) == 0.0
    assert abs(float(pe.scale.grad) - float((w * sinusoid_pe(12)[:30, 0].unsqueeze(0)).sum())) < 1e-4


# ---------------------------------------------------------------------------------------------------
def _model(sd, cfg=TINY_FP):
    from forwardtacotron_amd.fastpitch import FastPitch
    m = FastPitch(**cfg)
    m.load_state_dict(sd, strict=True)
    return m.cuda()


def test_state_dict_matches_reference_layout():
    from forwardtacotron_amd.fastpitch import FastPitch
    Z = load_npz('tiny_fastpitch.npz')
    ref_keys = [k[3:] for k in Z if k.startswith('sd/')]
    sd = FastPitch(**TINY_FP).state_dict()
    assert list(sd.keys()) == ref_keys
    for k in ref_keys:
        if not k.endswith('.pe'):
            assert tuple(sd[k].shape) == Z['sd/' + k].shape, k
        else:
            assert tuple(sd[k].shape) == (5000, 1, Z['sd/' + k].shape[-1])
            assert maxdiff(sd[k][:Z['sd/' + k].shape[0]], Z['sd/' + k]) == 0.0


def test_fastpitch_eval_train_and_generate():
    from forwardtacotron_amd import ops
    Z = load_npz('tiny_fastpitch.npz')
    m = _model(fp_state(Z, 'sd/'))
    batch = sub(Z, 'batch/')
    m.eval()
    with torch.no_grad():
        pred = m({k: v.clone().cuda() for k, v in batch.items()})
    for k in ('mel', 'mel_post', 'dur', 'pitch', 'energy'):
        assert pred[k].shape == Z['eval/' + k].shape, k
        assert maxdiff(pred[k].cpu(), Z['eval/' + k]) < 5e-5, k
    m.train()
    b = {k: v.clone().cuda() for k, v in batch.items()}
    pitch_t, energy_t = b['pitch'].clone(), b['energy'].clone()
    pred = m(b)
    c = TRAIN_CFG
    loss = ops.masked_l1(pred['mel'], b['mel'], b['mel_len']) + ops.masked_l1(pred['mel_post'], b['mel'], b['mel_len']) \
        + c['dur_loss_factor'] * ops.masked_l1(pred['dur'].unsqueeze(1), b['dur'].unsqueeze(1), b['x_len']) \
        + c['pitch_loss_factor'] * ops.masked_l1(pred['pitch'], pitch_t.unsqueeze(1), b['x_len']) \
        + c['energy_loss_factor'] * ops.masked_l1(pred['energy'], energy_t.unsqueeze(1), b['x_len'])
    assert abs(float(loss) - float(Z['loss/total'])) < 2e-5
    loss.backward()
    assert int(m.get_step()) == 1
    for k in ('mel', 'mel_post', 'dur', 'pitch', 'energy'):
        assert maxdiff(pred[k].detach().cpu(), Z['train/' + k]) < 5e-5, k
    worst, wk = 0.0, None
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        d = maxdiff(p.grad.cpu(), Z['grad/' + k])
        if d > worst:
            worst, wk = d, k
    assert worst < 1e-4, (worst, wk)
    m2 = _model(fp_state(Z, 'gen_sd/'))
    for tag, alpha in (('gen1', 0.9), ('gen2', 1.0)):
        out = m2.generate(torch.from_numpy(Z[tag + '/x']).cuda(), alpha=alpha)
        for k in ('mel', 'mel_post', 'dur', 'pitch', 'energy'):
            assert out[k].shape == Z[f'{tag}/{k}'].shape, (tag, k)
            assert maxdiff(out[k].float().cpu(), Z[f'{tag}/{k}']) < 5e-5, (tag, k)


def test_fastpitch_train_step_through_trainer():
    from forwardtacotron_amd.trainer import TrainStep
    Z = load_npz('tiny_fastpitch.npz')
    m = _model(fp_state(Z, 'sd/'))
    ts = TrainStep(m, lr=float(Z['lr']), train_cfg=TRAIN_CFG)
    out = ts.step({k: v.clone().cuda() for k, v in sub(Z, 'batch/').items()})
    assert abs(float(out['loss']) - float(Z['loss/total'])) < 2e-5
    assert abs(float(out['grad_norm']) - float(Z['grad_norm'])) < 1e-4 * max(1.0, float(Z['grad_norm']))
    sd = m.state_dict()
    grads = sub(Z, 'grad/')
    for k, v in sub(Z, 'sd_after/').items():
        if v.dtype.is_floating_point and not k.endswith('.pe'):
            live = grads[k].abs() > 1e-6          # rounding-noise gradients move by +-lr under Adam: excluded
            assert maxdiff(sd[k].cpu()[live], v[live]) < 3e-5, k


def test_fastpitch_mid_size_vs_oracle():
    """d_model 64 / heads 2 / conv 9+1 / 2 layers, ragged batch of 4: HIP model vs the CPU oracle (outputs, loss
    gradients)."""
    from oracle import fp_oracle as FP
    from oracle.ft_oracle import synthetic_batch
    from forwardtacotron_amd import ops
    cfg = dict(TINY_FP, durpred_d_model=32, durpred_d_fft=48, durpred_layers=2, pitch_d_model=32, pitch_n_heads=2,
               pitch_d_fft=40, energy_d_model=24, energy_n_heads=3, energy_d_fft=32, d_model=64, conv1_kernel=9,
               conv2_kernel=1, prenet_fft=96, prenet_heads=2, postnet_fft=128, postnet_heads=2, n_mels=20)
    from forwardtacotron_amd.fastpitch import FastPitch
    torch.manual_seed(11)
    m = FastPitch(**cfg)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.05 * torch.randn(p.shape))
    P = {k: v.clone() for k, v in m.state_dict().items()}
    batch = synthetic_batch(B=4, Tmax=23, n_mels=20, max_dur=6, seed=5)
    m = m.cuda().train()
    b = {k: v.clone().cuda() for k, v in batch.items()}
    pitch_t, energy_t = b['pitch'].clone(), b['energy'].clone()
    pred = m(b)
    c = TRAIN_CFG
    loss = ops.masked_l1(pred['mel'], b['mel'], b['mel_len']) + ops.masked_l1(pred['mel_post'], b['mel'], b['mel_len']) \
        + c['dur_loss_factor'] * ops.masked_l1(pred['dur'].unsqueeze(1), b['dur'].unsqueeze(1), b['x_len']) \
        + c['pitch_loss_factor'] * ops.masked_l1(pred['pitch'], pitch_t.unsqueeze(1), b['x_len']) \
        + c['energy_loss_factor'] * ops.masked_l1(pred['energy'], energy_t.unsqueeze(1), b['x_len'])
    loss.backward()
    _, _, info = FP.train_step(P, {}, batch, cfg, TRAIN_CFG, 1e-3, 1)
    assert abs(float(loss) - float(info['losses']['loss'])) < 5e-5
    for k in ('mel', 'dur', 'pitch', 'energy'):
        assert maxdiff(pred[k].detach().cpu(), info['pred'][k]) < 1e-4, k
    worst, wk = 0.0, None
    for k, p in m.named_parameters():
        d = maxdiff(p.grad.cpu(), info['grads'][k]) / max(1.0, float(info['grads'][k].abs().max()))
        if d > worst:
            worst, wk = d, k
    assert worst < 2e-4, (worst, wk)


def test_fastpitch_full_size_properties():
    """BASELINE configs[2] shape (fp32 arithmetic): singlespeaker.yaml FastPitch, bs=32, Tx=128, Tm=841.  No oracle at
    this size: the padding value is reproduced exactly beyond mel_len, mel_post is mel (fast_pitch.py:161-162), eval is
    deterministic, everything is finite, and one TrainStep moves every parameter by at most lr (first Adam step)."""
    from forwardtacotron_amd import data
    from forwardtacotron_amd.fastpitch import FastPitch
    from forwardtacotron_amd.trainer import TrainStep
    torch.manual_seed(0)
    cfg = dict(data.FASTPITCH_MODEL)
    m = FastPitch(**cfg).cuda().eval()
    batch = data.to_device(data.synthetic_batch(B=32, Tmax=128, n_mels=80, seed=0), 'cuda')
    Tm = int(batch['mel_len'].max())
    assert Tm == 841
    dur0 = batch['dur'].clone()
    with torch.no_grad():
        a = m(batch)
        batch['dur'].copy_(dur0)
        b = m(batch)
    for k in a:
        assert torch.equal(a[k], b[k]), k
        assert bool(torch.isfinite(a[k]).all()), k
    assert tuple(a['mel'].shape) == (32, 80, Tm + 1) and torch.equal(a['mel'], a['mel_post'])
    assert bool((a['mel'][:, :, Tm:] == -11.5129).all())
    for p_ in m.modules():                     # dropout at 0 for the |delta| <= lr property (masks rescale by 1/(1-p))
        if hasattr(p_, 'p'):
            p_.p = 0.0
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    ts = TrainStep(m, lr=1e-4, train_cfg=dict(data.SINGLESPEAKER_TRAIN))
    batch['dur'].copy_(dur0)
    out = ts.step(batch)
    ts.check()
    assert bool(torch.isfinite(out['loss'])) and float(out['grad_norm']) > 0
    worst = max(float((p.detach() - before[n]).abs().max()) for n, p in m.named_parameters())
    assert 0 < worst <= 1e-4 * 1.01


# ---------------------------------------------------------------------------------------------------
# BASELINE configs[2]: FastPitch with bf16 matmuls (operands rounded to bf16, fp32 accumulation).
# The reference has NO bf16 path (no autocast anywhere): bf16 parity is defined here, against the
# fp32 oracle at a stated looser tolerance, and is unpinned by the reference.
# ---------------------------------------------------------------------------------------------------
def test_bf16_gemm_rounds_operands_and_accumulates_in_fp32():
    """the bf16 kernels compute EXACTLY sum_k bf16(a) * bf16(b) in fp32 (to accumulation order): compare with a float64
    product of the rounded operands (tight) and with the unrounded product (bf16-sized error, not zero)"""
    from forwardtacotron_amd import hip as H
    torch.manual_seed(0)
    for rows, K, N in ((4096, 512, 384), (300, 64, 48), (26912, 80, 256)):
        x = torch.randn(rows, K, device='cuda')
        w = torch.randn(N, K, device='cuda')
        exact = H.linear_fwd(x, w)
        with H.gemm_precision('bf16'):
            got = H.linear_fwd(x, w)
            gdx = H.linear_bwd_data(got, w)
            gdw = H.linear_bwd_weight(got, x)
        assert H.set_gemm_precision('fp32') == 'fp32'                     # the context restored the default
        xr, wr = x.bfloat16().double(), w.bfloat16().double()
        ref = xr @ wr.t()
        scale = float(ref.abs().max())
        assert float((got.double() - ref).abs().max()) < 2e-6 * scale * max(1.0, (K / 64) ** 0.5), (rows, K, N)
        err = float((got - exact).abs().max())
        assert 1e-4 * scale < err < 3e-2 * scale, (rows, K, N, err / scale)  # genuinely bf16, not fp32
        g = got.bfloat16().double()
        assert float((gdx.double() - g @ wr).abs().max()) < 1e-5 * float((g @ wr).abs().max())
        assert float((gdw.double() - g.t() @ xr).abs().max()) < 1e-5 * float((g.t() @ xr).abs().max())


def _bf16_vs_oracle(cfg, B, Tmax, n_mels, seed, tol_mel, tol_loss):
    from oracle import fp_oracle as FP
    from oracle.ft_oracle import synthetic_batch
    from forwardtacotron_amd.fastpitch import FastPitch
    from forwardtacotron_amd.trainer import TrainStep
    torch.manual_seed(seed)
    m = FastPitch(**cfg)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.05 * torch.randn(p.shape))
    P = {k: v.clone() for k, v in m.state_dict().items()}
    batch = synthetic_batch(B=B, Tmax=Tmax, n_mels=n_mels, max_dur=6, seed=seed)
    newP, _, info = FP.train_step(P, {}, batch, cfg, TRAIN_CFG, 1e-3, 1)
    m = m.cuda()
    m.matmul_dtype = 'bf16'
    m.eval()
    with torch.no_grad():
        ev = m({k: v.clone().cuda() for k, v in batch.items()})
    m.matmul_dtype = 'fp32'
    with torch.no_grad():
        ev32 = m({k: v.clone().cuda() for k, v in batch.items()})
    d_bf = maxdiff(ev['mel'].cpu(), ev32['mel'].cpu())
    assert 1e-5 < d_bf < tol_mel, d_bf                     # bf16 really differs from fp32, within the stated tolerance
    m.matmul_dtype = 'bf16'
    ts = TrainStep(m, lr=1e-3, train_cfg=TRAIN_CFG)
    out = ts.step({k: v.clone().cuda() for k, v in batch.items()})
    ts.check()
    assert abs(float(out['loss']) - float(info['losses']['loss'])) < tol_loss * max(1.0, float(info['losses']['loss']))
    gn = float(info['grad_norm'])
    assert abs(float(out['grad_norm']) - gn) < 0.03 * max(1.0, gn)
    assert bool(all(torch.isfinite(p).all() for p in m.parameters()))
    return d_bf


def test_fastpitch_bf16_tiny_vs_fp32_oracle():
    _bf16_vs_oracle(TINY_FP, B=3, Tmax=9, n_mels=10, seed=3, tol_mel=5e-2, tol_loss=5e-3)


def test_fastpitch_bf16_mid_size_vs_fp32_oracle():
    cfg = dict(TINY_FP, durpred_d_model=32, durpred_d_fft=48, durpred_layers=2, pitch_d_model=32, pitch_n_heads=2,
               pitch_d_fft=40, energy_d_model=24, energy_n_heads=3, energy_d_fft=32, d_model=64, conv1_kernel=9,
               conv2_kernel=1, prenet_fft=96, prenet_heads=2, postnet_fft=128, postnet_heads=2, n_mels=20)
    _bf16_vs_oracle(cfg, B=4, Tmax=23, n_mels=20, seed=5, tol_mel=8e-2, tol_loss=5e-3)


# ---------------------------------------------------------------------------------------------------
# fused attention of the bf16 mode (csrc/ft_attn.hip): one flash-style kernel, backward by recomputation
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('B,T,d,nh,p_drop,ragged', [(2, 70, 128, 2, 0.0, True), (2, 70, 128, 2, 0.1, True),
                                                    (3, 200, 256, 2, 0.0, True), (2, 333, 256, 2, 0.1, True),
                                                    (1, 64, 256, 2, 0.0, False), (2, 129, 128, 2, 0.0, True),
                                                    (2, 841, 256, 2, 0.1, True)])
def test_fused_attention_vs_float64_on_the_rounded_operands(B, T, d, nh, p_drop, ragged):
    """ft_attn_fwd / ft_attn_bwd (QK^T -> masked softmax -> dropout -> PV in one kernel; no [B,h,T,T] tensor; backward
    recomputes from the saved log-sum-exp) against float64 attention (nn.MultiheadAttention's need_weights branch,
    common_layers.py:172-174, as oracle/fp_oracle.mha writes it out) on the SAME bf16-rounded q / k / v -- what is left is
    the bf16 rounding of the probabilities and fp32 accumulation: 1e-2 of the result's range.  The attention dropout is
    the library's counter-based mask (element index = flat index into [B,h,T,T]), reproduced here through ft_dropout on
    an index-shaped tensor, so the dropped variant is compared exactly too.  Head widths 64 and 128, ragged key padding,
    T not a multiple of the 64-key block / 128-query workgroup, the benchmark's T = 841."""
    import math
    from forwardtacotron_amd import hip as H
    torch.manual_seed(B * 1000 + T)
    hd = d // nh
    qkv = torch.randn(B, T, 3 * d, device='cuda') * 0.7
    lens = torch.randint(max(1, T // 2), T + 1, (B,))
    lens[0] = T
    key_pad = (torch.arange(T)[None, :] >= lens[:, None]).to(torch.uint8).cuda() if ragged else None
    scale, seed = 1.0 / math.sqrt(hd), 1234567
    att, lse2 = H.attn_fwd(qkv, key_pad, nh, scale, p_drop, seed)
    datt = torch.randn(B, T, d, device='cuda')
    dqkv = H.attn_bwd(qkv, att, datt, key_pad, lse2, nh, scale, p_drop, seed)
    again = H.attn_bwd(qkv, att, datt, key_pad, lse2, nh, scale, p_drop, seed)
    assert torch.equal(dqkv, again)                                     # no atomics: bitwise reproducible
    q, k, v = [t.bfloat16().double().cpu().reshape(B, T, nh, hd).permute(0, 2, 1, 3).requires_grad_(True)
               for t in qkv.split(d, dim=-1)]
    s = (q @ k.transpose(-1, -2)) * scale
    if key_pad is not None:
        s = s.masked_fill(key_pad.bool().cpu()[:, None, None, :], float('-inf'))
    P = torch.softmax(s, dim=-1)
    if p_drop > 0:
        keep = (H.dropout(torch.ones(B * nh * T * T, device='cuda'), p_drop, seed) > 0).double().cpu().reshape(B, nh, T, T)
        assert 0.85 < float(keep.mean()) < 0.95
        P = P * keep / (1 - p_drop)
    o = (P @ v).permute(0, 2, 1, 3).reshape(B, T, d)
    (o * datt.double().cpu()).sum().backward()
    assert maxdiff(att.cpu(), o.detach()) < 1e-2 * float(o.detach().abs().max())
    want = [t.grad.permute(0, 2, 1, 3).reshape(B, T, d) for t in (q, k, v)]
    for got, ref, name in zip(dqkv.cpu().split(d, dim=-1), want, 'qkv'):
        assert maxdiff(got, ref) < 1e-2 * float(ref.abs().max()), name
    if key_pad is not None:                                             # a padded key gets no gradient at all
        pad = key_pad.bool().cpu()
        assert float(dqkv.cpu()[:, :, d:][pad].abs().max() if pad.any() else 0.0) == 0.0


def test_bf16_attention_fused_equals_the_five_launch_form(monkeypatch):
    """MHAFn in bf16 mode, fused (default) against FT_ATTN_FUSED=0 (QK^T / softmax / PV and the four gradient products
    as separate launches with the [B,h,T,T] tensors in memory): outputs and every gradient agree to bf16 rounding."""
    from forwardtacotron_amd import hip as H
    from forwardtacotron_amd.fastpitch import MHAFn
    torch.manual_seed(5)
    B, T, d, nh = 3, 150, 256, 2
    x = torch.randn(B, T, d)
    in_w, in_b = torch.randn(3 * d, d) / d ** 0.5, torch.randn(3 * d) * 0.1
    out_w, out_b = torch.randn(d, d) / d ** 0.5, torch.randn(d) * 0.1
    lens = torch.tensor([150, 97, 120])
    key_pad = (torch.arange(T)[None, :] >= lens[:, None]).to(torch.uint8).cuda()
    w = torch.randn(B, T, d).cuda()
    res = {}
    with H.gemm_precision('bf16'):
        for fused in ('1', '0'):
            monkeypatch.setenv('FT_ATTN_FUSED', fused)
            ps = [t.clone().cuda().requires_grad_(True) for t in (x, in_w, in_b, out_w, out_b)]
            y = MHAFn.apply(ps[0], key_pad, ps[1], ps[2], ps[3], ps[4], nh, 0.1, 99)
            (y * w).sum().backward()
            res[fused] = [y.detach().cpu()] + [p.grad.cpu() for p in ps]
    for a, b_ in zip(res['1'], res['0']):
        assert maxdiff(a, b_) < 2e-2 * float(b_.abs().max())
