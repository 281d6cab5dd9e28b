"""GPU parity of the drop-in module (forward, losses, every gradient) against
  (a) golden fixtures captured from the imported reference (tests/golden/*.npz) and
  (b) the CPU oracle on seeded inputs, including deliberately awkward layer sizes.
Tolerances: fp32 forward 1e-4 abs on mel (north_star), tighter where the graph is short."""
import numpy as np
import pytest
import torch

from helpers import TINY, ODD, TRAIN_CFG, load_npz, sub, maxdiff

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ft():
    import forwardtacotron_amd as pkg
    from forwardtacotron_amd import model, ops, hip
    assert torch.cuda.is_available()
    return model, ops, hip


def cuda_batch(batch):
    return {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}


def load_sd(mod, sd):
    mod.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}, strict=True)
    return mod


# --------------------------------------------------------------------------------------------------- layers
@pytest.mark.parametrize('tag,relu', [('bnc_k5', True), ('bnc_k4', True), ('bnc_k3_norelu', False),
                                      ('bnc_k1', True), ('bnc_k2', True)])
def test_batchnorm_conv_golden(ft, tag, relu):
    model, ops, hip = ft
    L = load_npz('layers.npz')
    sd = sub(L, tag + '/sd/')
    Cout, Cin, k = sd['conv.weight'].shape
    m = load_sd(model.BatchNormConv(Cin, Cout, k, relu=relu), sd).cuda()
    x_bct = torch.from_numpy(L[tag + '/x'])
    x = x_bct.transpose(1, 2).contiguous().cuda()
    T = x.shape[1]
    m.eval()
    with torch.no_grad():
        y = m(x)
    assert maxdiff(y.cpu().transpose(1, 2), L[tag + '/eval'][:, :, :T]) < 1e-5
    m.train()
    xg = x.clone().requires_grad_(True)
    y = m(xg)
    ref = torch.from_numpy(L[tag + '/train'])
    assert maxdiff(y.detach().cpu().transpose(1, 2), ref[:, :, :T]) < 1e-5
    w = torch.from_numpy(L[tag + '/w'])
    if ref.shape[2] == T:      # odd k: full gradient check against the reference's autograd
        (y * w.transpose(1, 2).cuda()).sum().backward()
        assert maxdiff(xg.grad.cpu().transpose(1, 2), L[tag + '/dx']) < 2e-5
        assert maxdiff(m.conv.weight.grad.cpu(), L[tag + '/dW']) < 2e-5
        assert maxdiff(m.bnorm.weight.grad.cpu(), L[tag + '/dgamma']) < 2e-5
        assert maxdiff(m.bnorm.bias.grad.cpu(), L[tag + '/dbeta']) < 2e-5
    after = sub(L, tag + '/sd_after/')
    assert maxdiff(m.bnorm.running_mean.cpu(), after['bnorm.running_mean']) < 1e-6
    assert maxdiff(m.bnorm.running_var.cpu(), after['bnorm.running_var']) < 1e-6


def test_maxpool_highway_golden(ft):
    model, ops, hip = ft
    L = load_npz('layers.npz')
    x = torch.from_numpy(L['maxpool/x']).transpose(1, 2).contiguous().cuda()
    assert maxdiff(hip.maxpool2_fwd(x).cpu().transpose(1, 2), L['maxpool/y']) == 0.0
    h = load_sd(model.HighwayNetwork(6), sub(L, 'highway/sd/')).cuda()
    y = h(torch.from_numpy(L['highway/x']).cuda())
    assert maxdiff(y.detach().cpu(), L['highway/y']) < 2e-6


def test_maxpool_backward_first_max_wins(ft):
    model, ops, hip = ft
    # ties (post-ReLU zeros are common): torch routes the gradient to the FIRST maximal element
    x = torch.tensor([[[1.], [1.], [0.], [2.], [2.], [2.], [-1.]]])          # [1,7,1]
    mp = torch.nn.MaxPool1d(2, 1, 1)
    xr = x.transpose(1, 2).clone().requires_grad_(True)
    g = torch.arange(1., 8.).view(1, 1, 7)
    (mp(xr)[:, :, :7] * g).sum().backward()
    dx = hip.maxpool2_bwd(g.transpose(1, 2).contiguous().cuda(), x.cuda())
    assert maxdiff(dx.cpu().transpose(1, 2), xr.grad) == 0.0


def test_bigru_golden_and_grads(ft):
    model, ops, hip = ft
    from oracle import ft_oracle as O
    L = load_npz('layers.npz')
    sd = sub(L, 'gru/sd/')
    g = load_sd(model.GRU(5, 4), sd).cuda()
    x = torch.from_numpy(L['gru/x'])
    xg = x.cuda().requires_grad_(True)
    y = g(xg)
    assert maxdiff(y.detach().cpu(), L['gru/y']) < 2e-6
    w = torch.randn(y.shape, generator=torch.Generator().manual_seed(1))
    (y * w.cuda()).sum().backward()
    xo = x.double().requires_grad_(True)
    Po = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    (O.bigru(xo, Po, '') * w.double()).sum().backward()
    assert maxdiff(xg.grad.cpu(), xo.grad) < 2e-5
    for k in sd:
        assert maxdiff(getattr(g, k).grad.cpu(), Po[k].grad) < 2e-5, k


def test_bilstm_golden_and_grads(ft):
    model, ops, hip = ft
    from oracle import ft_oracle as O
    L = load_npz('layers.npz')
    sd = sub(L, 'lstm/sd/')
    m = load_sd(model.LSTM(5, 6), sd).cuda()
    x = torch.from_numpy(L['lstm/x'])
    lens = torch.from_numpy(L['lstm/lens'])
    xg = x.cuda().requires_grad_(True)
    y = m(xg, lens.cuda(), -11.5129)
    assert maxdiff(y.detach().cpu(), L['lstm/y_packed']) < 2e-6
    assert maxdiff(m(x.cuda(), None, 0.0).detach().cpu(), L['lstm/y_full']) < 2e-6
    w = torch.randn(y.shape, generator=torch.Generator().manual_seed(2))
    (y * w.cuda()).sum().backward()
    xo = x.double().requires_grad_(True)
    Po = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    (O.bilstm(xo, lens, Po, '') * w.double()).sum().backward()
    assert maxdiff(xg.grad.cpu(), xo.grad) < 2e-5
    for k in sd:
        assert maxdiff(getattr(m, k).grad.cpu(), Po[k].grad) < 2e-5, k


def test_cbhg_golden(ft):
    model, ops, hip = ft
    L = load_npz('layers.npz')
    c = load_sd(model.CBHG(K=4, in_channels=6, channels=8, proj_channels=[8, 6], num_highways=2, dropout=0.),
                sub(L, 'cbhg/sd/')).cuda()
    x = torch.from_numpy(L['cbhg/x']).transpose(1, 2).contiguous().cuda()
    c.eval()
    with torch.no_grad():
        assert maxdiff(c(x).cpu(), L['cbhg/eval']) < 1e-5
    c.train()
    assert maxdiff(c(x).detach().cpu(), L['cbhg/train']) < 1e-5


def test_series_predictor_golden(ft):
    model, ops, hip = ft
    L = load_npz('layers.npz')
    sp = load_sd(model.SeriesPredictor(num_chars=20, emb_dim=4, conv_dims=6, rnn_dims=3, dropout=0.),
                 sub(L, 'sp/sd/')).cuda()
    sp.eval()
    with torch.no_grad():
        y = sp(torch.from_numpy(L['sp/x']).cuda(), alpha=2.0)
    assert maxdiff(y.cpu(), L['sp/eval_alpha2']) < 1e-5


def test_masked_l1_golden(ft):
    model, ops, hip = ft
    L = load_npz('layers.npz')
    x = torch.from_numpy(L['l1/x']).cuda().requires_grad_(True)
    t = torch.from_numpy(L['l1/t']).cuda()
    lens = torch.from_numpy(L['l1/lens']).cuda()
    loss = ops.masked_l1(x, t, lens)
    assert abs(float(loss) - float(L['l1/loss'])) < 1e-6
    (loss * 3.0).backward()
    from oracle import ft_oracle as O
    xo = torch.from_numpy(L['l1/x']).requires_grad_(True)
    (O.masked_l1(xo, torch.from_numpy(L['l1/t']), torch.from_numpy(L['l1/lens'])) * 3.0).backward()
    assert maxdiff(x.grad.cpu(), xo.grad) < 1e-7


# --------------------------------------------------------------------------------------------------- model
def _train_step_hip(model_mod, ops, m, batch, train_cfg):
    m.train()
    b = cuda_batch({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
    pitch_t = b['pitch'].detach().clone()
    energy_t = b['energy'].detach().clone()
    pred = m(b)
    m1 = ops.masked_l1(pred['mel'], b['mel'], b['mel_len'])
    m2 = ops.masked_l1(pred['mel_post'], b['mel'], b['mel_len'])
    dl = ops.masked_l1(pred['dur'].unsqueeze(1), b['dur'].unsqueeze(1), b['x_len'])
    pl = ops.masked_l1(pred['pitch'], pitch_t.unsqueeze(1), b['x_len'])
    el = ops.masked_l1(pred['energy'], energy_t.unsqueeze(1), b['x_len'])
    loss = m1 + m2 + train_cfg['dur_loss_factor'] * dl + train_cfg['pitch_loss_factor'] * pl \
        + train_cfg['energy_loss_factor'] * el
    for p in m.parameters():
        p.grad = None
    loss.backward()
    return pred, {'loss': loss, 'mel': m1, 'mel_post': m2, 'dur': dl, 'pitch': pl, 'energy': el}, b


def test_tiny_model_golden_eval_and_train(ft):
    model, ops, hip = ft
    M = load_npz('tiny_model.npz')
    m = load_sd(model.ForwardTacotron(**TINY), sub(M, 'sd/')).cuda()
    batch = sub(M, 'batch/')
    m.eval()
    with torch.no_grad():
        pred = m(cuda_batch({k: v.clone() for k, v in batch.items()}))
    for k in ('mel', 'mel_post', 'dur', 'pitch', 'energy'):
        assert pred[k].shape == M['eval/' + k].shape, k
        assert maxdiff(pred[k].cpu(), M['eval/' + k]) < 5e-5, k
    pred, L, b = _train_step_hip(model, ops, m, batch, TRAIN_CFG)
    for k in ('mel', 'mel_post', 'dur', 'pitch', 'energy'):
        assert maxdiff(pred[k].detach().cpu(), M['train/' + k]) < 5e-5, k
    for k, gk in (('loss', 'total'), ('mel', 'mel'), ('mel_post', 'mel_post'), ('dur', 'dur'),
                  ('pitch', 'pitch'), ('energy', 'energy')):
        assert abs(float(L[k]) - float(M['loss/' + gk])) < 2e-5, k
    worst, worst_k = 0.0, None
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        d = maxdiff(p.grad.cpu(), M['grad/' + k])
        if d > worst:
            worst, worst_k = d, k
    assert worst < 1e-4, (worst, worst_k)
    # buffers after the training forward: BN running stats, counters, step
    after = sub(M, 'sd_after/')
    sd = m.state_dict()
    for k, v in after.items():
        if 'running_' in k:
            assert maxdiff(sd[k].cpu(), v) < 1e-5, k
        elif k.endswith('num_batches_tracked') or k == 'step':
            assert int(sd[k].reshape(-1)[0]) == int(v.reshape(-1)[0]), k
    assert np.array_equal(b['dur'].cpu().numpy(), np.maximum(M['batch/dur'], 0))   # in-place clamp side effect


@pytest.mark.parametrize('cfg_name', ['tiny', 'odd'])
def test_model_vs_oracle_train(ft, cfg_name):
    model, ops, hip = ft
    from oracle import ft_oracle as O
    cfg = TINY if cfg_name == 'tiny' else ODD
    torch.manual_seed(11)
    m = model.ForwardTacotron(**cfg)
    g = torch.Generator().manual_seed(5)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.weight.data = 1 + 0.2 * torch.randn(mod.weight.shape, generator=g)
            mod.bias.data = 0.1 * torch.randn(mod.bias.shape, generator=g)
    P = {k: v.clone() for k, v in m.state_dict().items()}
    batch = O.synthetic_batch(B=5, Tmax=13, n_mels=cfg['n_mels'], max_dur=5, seed=3)
    batch['dur'][1, 2] = -1.0          # negative duration: clamped in place by the LengthRegulator
    batch['dur'][2, :3] = 0.0
    # recompute mel_len/mel so that sum(dur) == mel_len still holds after the edits
    r = torch.from_numpy(O.lr_repeats(batch['dur'].numpy()))
    batch['mel_len'] = r.sum(1)
    Tm = int(batch['mel_len'].max())
    mel = torch.full((5, cfg['n_mels'], Tm + 1), -11.5129)
    for b in range(5):
        n = int(batch['mel_len'][b])
        mel[b, :, :n] = torch.randn(cfg['n_mels'], n, generator=g) * 2 - 5
    batch['mel'] = mel
    newP, _, info = O.train_step(P, {}, {k: v.clone() for k, v in batch.items()}, cfg, TRAIN_CFG, 1e-3, 1)
    m = m.cuda()
    pred, L, _ = _train_step_hip(model, ops, m, batch, TRAIN_CFG)
    for k in ('mel', 'mel_post', 'dur', 'pitch', 'energy'):
        assert maxdiff(pred[k].detach().cpu(), info['pred'][k]) < 5e-5, k
    assert abs(float(L['loss']) - float(info['losses']['loss'])) < 2e-5
    worst, worst_k = 0.0, None
    for k, p in m.named_parameters():
        d = maxdiff(p.grad.cpu(), info['grads'][k])
        if d > worst:
            worst, worst_k = d, k
    assert worst < 1e-4, (worst, worst_k)
    sd = m.state_dict()
    for k in sd:
        if 'running_' in k:
            assert maxdiff(sd[k].cpu(), newP[k]) < 1e-5, k


def test_generate_golden(ft):
    model, ops, hip = ft
    G = load_npz('generate.npz')
    m = load_sd(model.ForwardTacotron(**TINY), sub(G, 'sd/')).cuda()
    for tag in ('b1', 'b2'):
        out = m.generate(torch.from_numpy(G[tag + '/x']).cuda(), alpha=0.9)
        for k in ('mel', 'mel_post', 'dur', 'pitch', 'energy'):
            assert out[k].shape == G[f'{tag}/{k}'].shape, (tag, k)
            assert maxdiff(out[k].cpu(), G[f'{tag}/{k}']) < 5e-5, (tag, k)


def test_reference_unit_test_shapes(ft):
    """tests/test_forward_tacotron.py:19-46 of the reference, on the full singlespeaker config."""
    model, ops, hip = ft
    from helpers import FULL
    torch.manual_seed(0)
    m = model.ForwardTacotron(**FULL).cuda()
    batch = {
        'dur': torch.full((2, 10), fill_value=2).float(),
        'mel': torch.ones((2, 80, 20)).float(),
        'x': torch.ones((2, 10)).long(),
        'mel_len': torch.full((2,), fill_value=20).long(),
        'pitch': torch.ones((2, 10)).float(),
        'energy': torch.ones((2, 10)).float(),
    }
    pred = m(cuda_batch(batch))
    assert set(pred.keys()) == {'mel', 'mel_post', 'dur', 'pitch', 'energy'}
    assert tuple(pred['mel_post'].shape) == (2, 80, 20)
    assert tuple(pred['dur'].shape) == (2, 10)
    assert tuple(pred['pitch'].shape) == (2, 1, 10)
    assert tuple(pred['energy'].shape) == (2, 1, 10)
    gen = m.generate(x=torch.ones((1, 10)).long().cuda())
    assert gen['mel_post'].size(1) == 80 and tuple(gen['dur'].shape) == (1, 10)
    assert tuple(gen['pitch'].shape) == (1, 1, 10)


def test_long_form_eval_vs_oracle(ft):
    """BASELINE configs[4] in miniature: long items (160 tokens -> ~900 frames, ragged) through the eval /
    mel-generation path; the recurrences run 900 dependent persistent steps.  HIP vs oracle on every output."""
    model, ops, hip = ft
    from oracle import ft_oracle as O
    cfg = dict(TINY, rnn_dims=32, postnet_dims=16)              # H multiples of 16: persistent kernels
    torch.manual_seed(21)
    m = model.ForwardTacotron(**cfg).eval()
    P = {k: v.clone() for k, v in m.state_dict().items()}
    batch = O.synthetic_batch(B=3, Tmax=160, n_mels=cfg['n_mels'], max_dur=12, seed=8)
    with torch.no_grad():
        want, _ = O.forward(P, {k: v.clone() for k, v in batch.items()}, cfg, training=False)
        got = m.cuda()(cuda_batch(batch))
    hip.check_rnn_status()
    assert int(batch['mel_len'].max()) > 800
    for k in ('mel', 'mel_post', 'dur', 'pitch', 'energy'):
        assert got[k].shape == want[k].shape, k
        assert maxdiff(got[k].cpu(), want[k]) < 1e-4, k


def test_long_form_full_shape(ft):
    """BASELINE configs[4] at its FULL shape: singlespeaker.yaml model, B=128 items x 1000 token slots (x_len 500..1000),
    explicit durations ~ randint(1,12) through _generate_mel (forward_tacotron.py:205-234; an untrained duration
    predictor degenerates to the fill_(2.) fallback) -- ~575 k frames, a [128,~6000,512] LengthRegulator
    output, a 780 k-row conv bank.  Checks: the LengthRegulator's rows are bit-exact copies of their source tokens
    (want_src index map vs the numpy oracle map); every output finite; and TWO items -- the longest and a short one,
    which sits beside ~2500 zero-padded frames -- equal the oracle's _generate_mel run on that item ALONE, zero-padded
    to the batch's frame count (eval-mode BatchNorm is per-position, so batch items are independent)."""
    model, ops, hip = ft
    from forwardtacotron_amd import data
    from oracle import ft_oracle as O
    cfg = dict(data.SINGLESPEAKER_MODEL)
    torch.manual_seed(0)
    m = model.ForwardTacotron(**cfg).eval()
    P = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.cuda()
    batch = data.synthetic_batch(B=128, Tmax=1000, n_mels=80, seed=0)
    x, dur = batch['x'], batch['dur']
    pitch, energy = batch['pitch'].unsqueeze(1), batch['energy'].unsqueeze(1)
    total = batch['mel_len']
    Tm = int(total.max())
    assert int(total.sum()) > 500_000 and Tm > 5500
    torch.cuda.reset_peak_memory_stats()
    with torch.no_grad():
        out = m._generate_mel(x.cuda(), dur.clone().cuda(), pitch.cuda(), energy.cuda())
    torch.cuda.synchronize()
    hip.check_rnn_status()
    assert tuple(out['mel'].shape) == (128, 80, Tm) and tuple(out['mel_post'].shape) == (128, 80, Tm)
    assert bool(torch.isfinite(out['mel']).all()) and bool(torch.isfinite(out['mel_post']).all())
    # LengthRegulator index map at the full shape, bit-exact rows
    src_want, tot_want = O.lr_index_map(dur.numpy())
    cum, tot = hip.lr_scan(dur.clone().cuda())
    assert np.array_equal(tot.cpu().numpy(), tot_want)
    rows = torch.randn(128, 1000, 64, device='cuda')
    y, src = hip.lr_expand(rows, cum, Tm, want_src=True)
    assert np.array_equal(src.cpu().numpy(), src_want)
    bi = torch.arange(128, device='cuda').unsqueeze(1).expand(128, Tm)
    valid = src >= 0
    assert torch.equal(y[valid], rows[bi[valid], src[valid].long()]) and float(y[~valid].abs().sum()) == 0.0
    del rows, y
    # two items against the oracle run on each alone (padded to the batch's frame count)
    for b in (int(total.argmax()), int(total.argmin()), 17):
        want = O.generate_mel(P, x[b:b + 1], dur[b:b + 1].clone(), pitch[b:b + 1], energy[b:b + 1], cfg,
                              pad_frames_to=Tm)
        n = int(total[b])
        for k in ('mel', 'mel_post'):
            assert want[k].shape[-1] == Tm
            assert maxdiff(out[k][b, :, :n].cpu(), want[k][0, :, :n]) < 1e-4, (k, b)
            assert maxdiff(out[k][b].cpu(), want[k][0]) < 1e-4, (k, b, 'padded frames')
    assert torch.cuda.max_memory_allocated() < 80 * 2 ** 30


def test_full_size_properties(ft):
    """The benchmark configuration itself (singlespeaker.yaml, bs=32, Tx=128, Tm=841): no oracle at this size, so the
    size-independent properties -- padding value beyond mel_len is reproduced exactly, LengthRegulator output rows are
    bit-exact copies, eval is deterministic, and the train step moves every parameter by at most lr (Adam, step 1)."""
    model, ops, hip = ft
    from forwardtacotron_amd import data
    from forwardtacotron_amd.trainer import TrainStep
    from forwardtacotron_amd import hip as H
    torch.manual_seed(0)
    m = model.ForwardTacotron(**data.SINGLESPEAKER_MODEL).cuda().eval()
    batch = data.to_device(data.synthetic_batch(B=32, Tmax=128, n_mels=80, seed=0), 'cuda')
    dur0 = batch['dur'].clone()
    with torch.no_grad():
        a = m(batch)
        batch['dur'].copy_(dur0)
        b = m(batch)
    hip.check_rnn_status()
    Tm = int(batch['mel_len'].max())
    assert tuple(a['mel'].shape) == (32, 80, Tm + 1) and Tm == 841
    for k in a:
        assert torch.equal(a[k], b[k]), k                                 # deterministic
    assert bool((a['mel'][:, :, Tm:] == -11.5129).all()) and bool((a['mel_post'][:, :, Tm:] == -11.5129).all())
    assert bool(torch.isfinite(a['mel_post']).all())
    # LengthRegulator: frame f of item b is an exact copy of the token row it came from
    x = torch.randn(32, 128, 512, device='cuda')
    cum, total = H.lr_scan(dur0.clone())
    y, src = H.lr_expand(x, cum, Tm, want_src=True)
    bi = torch.arange(32, device='cuda').unsqueeze(1).expand(32, Tm)
    valid = src >= 0
    assert int(valid.sum()) == int(batch['mel_len'].sum()) == 19320
    assert torch.equal(y[valid], x[bi[valid], src[valid].long()]) and float(y[~valid].abs().sum()) == 0.0
    # one optimisation step: |delta| <= lr for every parameter (first Adam step), loss finite
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    ts = TrainStep(m, lr=1e-4, train_cfg=dict(data.SINGLESPEAKER_TRAIN))
    batch['dur'].copy_(dur0)
    out = ts.step(batch)
    hip.check_rnn_status()
    assert bool(torch.isfinite(out['loss'])) and float(out['grad_norm']) > 0
    worst = max(float((p.detach() - before[n]).abs().max()) for n, p in m.named_parameters())
    assert 0 < worst <= 1e-4 * 1.01          # lr, plus the fp32 rounding of p - delta for |p| up to ~1


def test_expanded_length_beyond_mel_len_golden(ft):
    """tests/golden/tiny_crop.npz (captured from the reference): rounded durations expand to T_lr = max(mel_len) + 5;
    pad_packed_sequence returns max(mel_len) frames, so lin / the postnet's BatchNorm statistics / GRU run on that many.
    And the error case: an item packed with more frames than its durations give raises like the reference's LSTM."""
    model, ops, hip = ft
    M, C = load_npz('tiny_model.npz'), load_npz('tiny_crop.npz')
    m = load_sd(model.ForwardTacotron(**TINY), sub(M, 'sd/')).cuda()
    batch = sub(C, 'batch/')
    pred, L, _ = _train_step_hip(model, ops, m, batch, TRAIN_CFG)
    for k in ('mel', 'mel_post', 'dur', 'pitch', 'energy'):
        assert pred[k].shape == C['train/' + k].shape, k
        assert maxdiff(pred[k].detach().cpu(), C['train/' + k]) < 5e-5, k
    assert abs(float(L['loss']) - float(C['loss/total'])) < 2e-5
    worst = max(maxdiff(p.grad.cpu(), C['grad/' + k]) for k, p in m.named_parameters())
    assert worst < 1e-4, worst
    sd = m.state_dict()
    for k, v in sub(C, 'sd_after/').items():
        assert maxdiff(sd[k].cpu(), v) < 1e-5, k
    bad = sub(M, 'batch/')
    bad['mel_len'] = bad['mel_len'].clone()
    bad['mel_len'][int(bad['mel_len'].argmax())] += 2
    with pytest.raises(hip._lib.FtError, match='packed'):
        m(cuda_batch(bad))


def test_generate_jit_beta_scales_the_pitch(ft):
    """forward_tacotron.py:186-200: generate_jit(x, alpha, beta) == generate(x, alpha, pitch_function = p * beta);
    checked against generate() of this module AND against the oracle's generate with the same pitch function."""
    model, ops, hip = ft
    from oracle import ft_oracle as O
    G = load_npz('generate.npz')
    sd = sub(G, 'sd/')
    m = load_sd(model.ForwardTacotron(**TINY), sd).cuda().eval()
    x = torch.from_numpy(G['b2/x'])
    for alpha, beta in ((1.0, 1.0), (0.9, 1.7), (1.3, 0.4)):
        got = m.generate_jit(x.cuda(), alpha, beta)
        ref = m.generate(x.cuda(), alpha=alpha, pitch_function=lambda p: p * beta)
        want = O.generate(sd, x, TINY, alpha=alpha, pitch_function=lambda p: p * beta)
        for k in ('mel', 'mel_post', 'dur', 'pitch', 'energy'):
            assert got[k].shape == want[k].shape, (k, alpha, beta)
            assert maxdiff(got[k].cpu(), ref[k].cpu()) < 1e-6, (k, alpha, beta)
            assert maxdiff(got[k].cpu(), want[k]) < 5e-5, (k, alpha, beta)
    base = m.generate_jit(x.cuda(), 1.0, 1.0)
    assert maxdiff(m.generate_jit(x.cuda(), 1.0, 2.0)['pitch'].cpu(), 2.0 * base['pitch'].cpu()) < 1e-6


def test_torchscript_generate_jit(ft, tmp_path):
    """README.md:159-171 of the reference on the drop-in: torch.jit.script(model).generate_jit on the GPU, before and
    after torch.jit.save / load, equals the eager generate_jit and the reference's generate() golden (beta = 1)."""
    model, ops, hip = ft
    G = load_npz('generate.npz')
    m = load_sd(model.ForwardTacotron(**TINY), sub(G, 'sd/')).cuda().eval()
    s = torch.jit.script(m)
    path = str(tmp_path / 'ft_script.pt')
    s.save(path)
    loaded = torch.jit.load(path, map_location='cuda')
    x = torch.from_numpy(G['b2/x']).cuda()
    for mod in (s, loaded):
        out = mod.generate_jit(x, 0.9, 1.0)
        for k in ('mel', 'mel_post', 'dur', 'pitch', 'energy'):
            assert maxdiff(out[k].cpu(), G[f'b2/{k}']) < 5e-5, k
        a, b = mod.generate_jit(x, 1.2, 0.5), m.generate_jit(x, 1.2, 0.5)
        for k in a:
            assert torch.equal(a[k], b[k]), k
        assert set(mod(x).keys()) == {'mel', 'mel_post', 'dur', 'pitch', 'energy'}


def test_eval_forward_with_autograd_enabled_is_refused(ft):
    """the eval path records no autograd graph (BatchNorm folded into the conv epilogue): asked for gradients it must
    raise, not return tensors that silently carry none"""
    model, ops, hip = ft
    M = load_npz('tiny_model.npz')
    m = load_sd(model.ForwardTacotron(**TINY), sub(M, 'sd/')).cuda().eval()
    batch = cuda_batch(sub(M, 'batch/'))
    with pytest.raises(hip._lib.FtError, match='not differentiable'):
        m(batch)
    with torch.no_grad():
        m(cuda_batch(sub(M, 'batch/')))


def test_cpu_tensors_are_refused(ft):
    model, ops, hip = ft
    m = model.ForwardTacotron(**TINY)
    with pytest.raises(Exception):
        m({'x': torch.ones(1, 3).long(), 'mel': torch.ones(1, 10, 4), 'dur': torch.ones(1, 3),
           'mel_len': torch.tensor([3]), 'pitch': torch.ones(1, 3), 'energy': torch.ones(1, 3)})
