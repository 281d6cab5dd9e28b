"""GPU parity AT THE BENCHMARK CONFIGURATION and at production widths ("Next round" 1).

(a) BASELINE configs[1] itself -- singlespeaker.yaml widths, bs=32, Tx=128, Tm=841, seed-0 batch, all dropout 0: one
    HIP train step against oracle/ft_torch_cpu.CpuTrainer (the stock-op restatement, pinned to the reference's goldens
    and -- at this very width -- to the live import, tests/test_refimport_full_size.py): mel / mel_post <= 1e-4 abs (the
    north-star bar), the five loss terms, the gradient norm, EVERY parameter's gradient, the BatchNorm running statistics.
    Follows models/forward_tacotron.py:118-165 and trainer/forward_trainer.py:73-99 of the reference.
(b) The decoder LSTM at its production width (H=512, B=32, input 512) in its DEFAULT persistent templates (4-wave forward,
    reduce-scatter BPTT at G*H = 2048) and in the per-step kernels, each against the float64 oracle; and the GRU-256 at the
    un-tamed 0.3 weight scale of the narrower cases, where round 2 saw persistent and per-step drift 1.8e-4 apart and
    scaled the weights down: both forms are held against float64 together with a THIRD fp32 implementation (stock torch
    CPU) -- if all three sit equally far from float64 the recurrence is chaotic at that scale, if one of ours is farther
    there is a bug.
(c) FastPitch at d_model=256 / 2 heads (head_dim 128) with > 4096 frame rows against oracle/fp_oracle (fp32), i.e. through
    the 128x128 split-GEMM tiles inside a model (models/fast_pitch.py:123-165), and the bf16 mode at the full configs[2]
    shape as a property test (finite, padding exact, |delta p| <= lr): the reference has no bf16 path, nothing to pin it to.
"""
import os

import pytest
import torch

from helpers import TRAIN_CFG, maxdiff

pytestmark = pytest.mark.gpu


def _set_persistent(flag):
    from forwardtacotron_amd import _lib
    return _lib.lib().ft_rnn_set_persistent(int(flag))


# ---------------------------------------------------------------------------------------------------
# (a) the benchmark configuration, train mode, against the pinned CPU restatement
# ---------------------------------------------------------------------------------------------------
def test_benchmark_config_train_step_vs_cpu_oracle():
    from forwardtacotron_amd import data, hip as H
    from forwardtacotron_amd.model import ForwardTacotron
    from forwardtacotron_amd.trainer import TrainStep
    from oracle import ft_torch_cpu as C
    cfg = dict(data.SINGLESPEAKER_MODEL, durpred_dropout=0.0, pitch_dropout=0.0, energy_dropout=0.0,
               prenet_dropout=0.0, postnet_dropout=0.0)
    tc = dict(data.SINGLESPEAKER_TRAIN)
    lr = 1e-4
    torch.manual_seed(0)
    m = ForwardTacotron(**cfg)
    P = {k: v.clone() for k, v in m.state_dict().items()}
    batch = data.synthetic_batch(B=32, Tmax=128, n_mels=80, seed=0)
    assert int(batch['mel_len'].max()) == 841 and int(batch['mel_len'].sum()) == 19320

    # HIP step first (seconds), then the CPU step (~45 s on the box's 16-thread share)
    m = m.cuda()
    ts = TrainStep(m, lr=lr, train_cfg=tc)
    seen = {}
    inner = ts.losses

    def losses(pred, *a, **kw):
        seen.update({k: v.detach() for k, v in pred.items()})
        return inner(pred, *a, **kw)

    ts.losses = losses
    out = ts.step({k: v.clone().cuda() for k, v in batch.items()})
    ts.check()
    H.check_rnn_status()
    grads = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}
    pred = {k: v.cpu() for k, v in seen.items()}
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    out = {k: float(v) for k, v in out.items()}

    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cpu = C.CpuTrainer(P, cfg, tc, lr)
    info = cpu.step(batch)
    after = cpu.state_dict()

    # forward: the north-star bar
    report = {}
    for k in ('mel', 'mel_post'):
        assert pred[k].shape == info['pred'][k].shape == (32, 80, 842), k
        d = report[k] = maxdiff(pred[k], info['pred'][k])
        assert d <= 1e-4, (k, d)
    for k in ('dur', 'pitch', 'energy'):
        d = report[k] = maxdiff(pred[k].reshape(-1), info['pred'][k].reshape(-1))
        assert d <= 1e-4, (k, d)
    # losses and the gradient norm
    for k in ('mel', 'mel_post', 'dur', 'pitch', 'energy', 'loss'):
        want = float(info['losses'][k])
        assert abs(out[k] - want) <= 2e-5 * max(1.0, abs(want)), (k, out[k], want)
    gn = float(info['grad_norm'])
    assert abs(out['grad_norm'] - gn) <= 1e-4 * max(1.0, gn), (out['grad_norm'], gn)
    # every parameter's gradient
    worst, wk = 0.0, None
    assert set(grads) == set(info['grads'])
    for k, g in info['grads'].items():
        d = maxdiff(grads[k], g) / max(1.0, float(g.abs().max()))
        if d > worst:
            worst, wk = d, k
    assert worst <= 1e-4, (worst, wk)
    report['grad (rel. to max(1,|g|inf))'] = (worst, wk)
    report['grad_norm'] = (out['grad_norm'], gn)
    print('benchmark-config parity, max |HIP - CPU oracle|:', report)
    # BatchNorm running statistics and the counters
    for k, v in after.items():
        if 'running_' in k:
            d = maxdiff(sd[k], v) / max(1.0, float(v.abs().max()))
            assert d <= 1e-5, (k, d)
        elif k.endswith('num_batches_tracked') or k == 'step':
            assert int(sd[k].reshape(-1)[0]) == int(v.reshape(-1)[0]) == 1, k
    # post-Adam parameters where the gradient is not rounding noise (Adam's first step moves by lr * sign(g))
    worst, wk = 0.0, None
    for k, g in info['grads'].items():
        live = g.abs() > 1e-6
        if bool(live.any()):
            d = float((sd[k] - after[k]).abs()[live].max())
            if d > worst:
                worst, wk = d, k
    assert worst <= 0.1 * lr, (worst, wk)


# ---------------------------------------------------------------------------------------------------
# (b) recurrences at production width against float64
# ---------------------------------------------------------------------------------------------------
def _rnn_params(G, I, Hh, g, sh):
    P = {}
    for sfx in ('', '_reverse'):
        P['weight_ih_l0' + sfx] = torch.randn(G * Hh, I, generator=g) * (1.0 / I ** 0.5)
        P['weight_hh_l0' + sfx] = torch.randn(G * Hh, Hh, generator=g) * sh
        P['bias_ih_l0' + sfx] = torch.randn(G * Hh, generator=g) * 0.1
        P['bias_hh_l0' + sfx] = torch.randn(G * Hh, generator=g) * 0.1
    return P


def test_lstm_512_production_templates_vs_float64_oracle():
    """H=512, B=32, input width 512 (= 2 * prenet_dims), T=48, ragged packed lengths: the forward and the BPTT of the
    DEFAULT persistent launch (the 4-wave / 4-block forward template, ft_rnn_bwd_rs_kernel at G*H = 2048) and of the
    per-step kernels, each against oracle.bilstm in float64 -- the bars of test_lstm_persistent_vs_step_vs_oracle."""
    from forwardtacotron_amd import model, hip
    from oracle import ft_oracle as O
    B, T, I, Hh = 32, 48, 512, 512
    g = torch.Generator().manual_seed(512)
    P = _rnn_params(4, I, Hh, g, 1.0 / Hh ** 0.5)
    x = torch.randn(B, T, I, generator=g)
    w = torch.randn(B, T, 2 * Hh, generator=g)
    lens = torch.randint(1, T + 1, (B,), generator=g)
    lens[0], lens[-1], lens[5] = T, 1, T
    xo = x.double().requires_grad_(True)
    Po = {k: v.double().requires_grad_(True) for k, v in P.items()}
    yo = O.bilstm(xo, lens, Po, '')
    (yo * w.double()).sum().backward()
    for mode in (1, 0):
        old = _set_persistent(mode)
        try:
            c0 = hip.rnn_counters()
            m = model.LSTM(I, Hh)
            m.load_state_dict(P)
            m = m.cuda()
            xg = x.cuda().requires_grad_(True)
            y = m(xg, lens.cuda(), -11.5129)
            (y * w.cuda()).sum().backward()
            hip.check_rnn_status()
            c1 = hip.rnn_counters()
        finally:
            _set_persistent(old)
        assert (c1[0] - c0[0]) == (2 if mode else 0), 'forward + backward must run in the form under test'
        assert maxdiff(y.detach().cpu(), yo.detach()) < 2e-5, mode
        assert maxdiff(xg.grad.cpu(), xo.grad) < 1e-4 * max(1.0, float(xo.grad.abs().max())), mode
        for k in P:
            d = maxdiff(getattr(m, k).grad.cpu(), Po[k].grad)
            assert d < 2e-4 * max(1.0, float(Po[k].grad.abs().max())), (mode, k, d)


def test_gru_256_at_the_untamed_weight_scale_both_forms_vs_float64():
    """Round 2 scaled this case's recurrent weights by 1/sqrt(H) after persistent and per-step kernels differed by 1.8e-4
    at scale 0.3 (spectral radius ~ 4.8: a chaotic recurrence).  Shown here instead of argued: at the ORIGINAL scale the
    persistent form, the per-step form and stock torch CPU fp32 (a third, unrelated fp32 implementation) are each compared
    with the float64 oracle.  A healthy kernel is no farther from float64 than the other fp32 implementations are."""
    from forwardtacotron_amd import model, hip
    from oracle import ft_oracle as O
    from oracle import ft_torch_cpu as C
    B, T, I, Hh = 32, 21, 48, 256
    g = torch.Generator().manual_seed(B * 7 + Hh)
    P = _rnn_params(3, I, Hh, g, 0.3)
    for k in P:
        if 'weight_ih' in k:
            P[k] = P[k] * (0.3 * I ** 0.5)          # 0.3 like the round-2 case
    x = torch.randn(B, T, I, generator=g)
    w = torch.randn(B, T, 2 * Hh, generator=g)
    xo = x.double().requires_grad_(True)
    Po = {k: v.double().requires_grad_(True) for k, v in P.items()}
    yo = O.bigru(xo, Po, '')
    (yo * w.double()).sum().backward()
    dist = {}
    # third implementation: stock torch CPU fp32 (_VF.gru)
    xc = x.clone().requires_grad_(True)
    Pc = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    yc = C._bigru(xc, Pc, '', True)
    (yc * w).sum().backward()
    dist['torch_cpu'] = (maxdiff(yc.detach(), yo.detach()), maxdiff(xc.grad, xo.grad))
    for mode, name in ((1, 'persistent'), (0, 'per_step')):
        old = _set_persistent(mode)
        try:
            m = model.GRU(I, Hh)
            m.load_state_dict(P)
            m = m.cuda()
            xg = x.cuda().requires_grad_(True)
            y = m(xg)
            (y * w.cuda()).sum().backward()
            hip.check_rnn_status()
        finally:
            _set_persistent(old)
        dist[name] = (maxdiff(y.detach().cpu(), yo.detach()), maxdiff(xg.grad.cpu(), xo.grad))
    print('GRU-256 @0.3: max |fp32 - float64| (forward, dx):', dist)
    scale_dx = max(1.0, float(xo.grad.abs().max()))
    ref_f = max(dist['torch_cpu'][0], 1e-6)
    ref_b = max(dist['torch_cpu'][1], 1e-6 * scale_dx)
    for name in ('persistent', 'per_step'):
        # no farther from float64 than 4x the unrelated fp32 implementation (or than fp32 rounding noise itself)
        assert dist[name][0] <= max(4 * ref_f, 2e-5), (name, dist)
        assert dist[name][1] <= max(4 * ref_b, 1e-4 * scale_dx), (name, dist)
    # and neither of ours is an outlier against the other
    lo, hi = sorted((dist['persistent'][0], dist['per_step'][0]))
    assert hi <= max(4 * lo, 2e-5), dist


# ---------------------------------------------------------------------------------------------------
# (c) FastPitch at production head width against the fp32 oracle; bf16 at the full shape
# ---------------------------------------------------------------------------------------------------
def test_fastpitch_wide_vs_oracle():
    """d_model 256, 2 heads (head_dim 128), fft 1024, conv 9 / 1, predictors d=128: > 4096 frame rows, so the 128x128
    split-GEMM tiles (rows / weight-gradient forms) and the head_dim-128 attention products run inside a model."""
    from oracle import fp_oracle as FP
    from oracle.ft_oracle import synthetic_batch
    from forwardtacotron_amd import ops
    from forwardtacotron_amd.fastpitch import FastPitch
    from forwardtacotron_amd import data
    cfg = dict(data.FASTPITCH_MODEL, durpred_dropout=0.0, pitch_dropout=0.0, energy_dropout=0.0, prenet_dropout=0.0,
               postnet_dropout=0.0, durpred_layers=2, pitch_layers=2, energy_layers=2, prenet_layers=2, postnet_layers=2)
    torch.manual_seed(17)
    m = FastPitch(**cfg)
    P = {k: v.clone() for k, v in m.state_dict().items()}
    batch = synthetic_batch(B=10, Tmax=96, n_mels=80, max_dur=12, seed=6)
    assert int(batch['mel_len'].max()) * 10 > 4096
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
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    _, _, info = FP.train_step(P, {}, batch, cfg, TRAIN_CFG, 1e-3, 1)
    assert abs(float(loss.detach()) - float(info['losses']['loss'])) < 5e-5 * max(1.0, float(info['losses']['loss']))
    for k in ('mel', 'dur', 'pitch', 'energy'):
        assert maxdiff(pred[k].detach().cpu(), info['pred'][k]) < 1e-4, k
    worst, wk = 0.0, None
    for k, p in m.named_parameters():
        d = maxdiff(p.grad.cpu(), info['grads'][k]) / max(1.0, float(info['grads'][k].abs().max()))
        if d > worst:
            worst, wk = d, k
    assert worst < 2e-4, (worst, wk)


def test_fastpitch_bf16_full_size_properties():
    """BASELINE configs[2] as benchmarked (bf16 matmuls, bs=32, Tx=128, Tm=841).  The reference has no bf16 path: parity
    is unpinned by it, so this is the size-independent property set -- finite outputs, the padding value reproduced
    exactly, mel_post is mel, bf16 really differs from fp32 by a bf16-sized amount, one Adam step moves every
    parameter by at most lr."""
    from forwardtacotron_amd import data
    from forwardtacotron_amd.fastpitch import FastPitch
    from forwardtacotron_amd.trainer import TrainStep
    torch.manual_seed(0)
    m = FastPitch(**dict(data.FASTPITCH_MODEL)).cuda().eval()
    batch = data.to_device(data.synthetic_batch(B=32, Tmax=128, n_mels=80, seed=0), 'cuda')
    Tm = int(batch['mel_len'].max())
    dur0 = batch['dur'].clone()
    with torch.no_grad():
        a32 = m(batch)
        m.matmul_dtype = 'bf16'
        batch['dur'].copy_(dur0)
        a = m(batch)
        batch['dur'].copy_(dur0)
        b = m(batch)
    for k in a:
        assert torch.equal(a[k], b[k]), k
        assert bool(torch.isfinite(a[k]).all()), k
    assert tuple(a['mel'].shape) == (32, 80, Tm + 1) and torch.equal(a['mel'], a['mel_post'])
    assert bool((a['mel'][:, :, Tm:] == -11.5129).all())
    d = maxdiff(a['mel'].cpu(), a32['mel'].cpu())
    assert 1e-5 < d < 0.25, d
    for p_ in m.modules():
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
