"""GPU parity of the multispeaker variant vs goldens captured from the imported reference."""
import numpy as np
import pytest
import torch

from helpers import TINY_MULTI, TRAIN_CFG_MULTI, load_npz, sub, maxdiff

pytestmark = pytest.mark.gpu


def _model(sd):
    from forwardtacotron_amd.multi_model import MultiForwardTacotron
    m = MultiForwardTacotron(**TINY_MULTI)
    m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}, strict=True)
    return m.cuda()


def test_state_dict_matches_reference_layout():
    from forwardtacotron_amd.multi_model import MultiForwardTacotron
    M = load_npz('tiny_multi.npz')
    ref_keys = [k[3:] for k in M if k.startswith('sd/')]
    m = MultiForwardTacotron(**TINY_MULTI)
    sd = m.state_dict()
    assert list(sd.keys()) == ref_keys
    for k in ref_keys:
        assert tuple(sd[k].shape) == M['sd/' + k].shape, k


def test_multispeaker_eval_train_and_generate():
    from forwardtacotron_amd import ops, hip
    M = load_npz('tiny_multi.npz')
    m = _model(sub(M, 'sd/'))
    batch = sub(M, 'batch/')
    m.eval()
    with torch.no_grad():
        pred = m({k: v.clone().cuda() for k, v in batch.items()})
    for k in ('mel', 'mel_post', 'dur', 'pitch', 'energy', 'pitch_cond'):
        assert pred[k].shape == M['eval/' + k].shape, k
        assert maxdiff(pred[k].cpu(), M['eval/' + k]) < 5e-5, k
    m.train()
    b = {k: v.clone().cuda() for k, v in batch.items()}
    pitch_t, energy_t = b['pitch'].clone(), b['energy'].clone()
    pred = m(b)
    c = TRAIN_CFG_MULTI
    loss = ops.masked_l1(pred['mel'], b['mel'], b['mel_len']) + ops.masked_l1(pred['mel_post'], b['mel'], b['mel_len']) \
        + c['dur_loss_factor'] * ops.masked_l1(pred['dur'].unsqueeze(1), b['dur'].unsqueeze(1), b['x_len']) \
        + c['pitch_loss_factor'] * ops.masked_l1(pred['pitch'], pitch_t.unsqueeze(1), b['x_len']) \
        + c['energy_loss_factor'] * ops.masked_l1(pred['energy'], energy_t.unsqueeze(1), b['x_len'])
    ce = ops.cross_entropy(pred['pitch_cond'], b['pitch_cond'], 0)
    assert abs(float(ce) - float(M['loss/pitch_cond'])) < 1e-5
    loss = loss + c['pitch_cond_loss_factor'] * ce
    assert abs(float(loss) - float(M['loss/total'])) < 2e-5
    loss.backward()
    hip.check_rnn_status()
    for k in ('mel', 'mel_post', 'dur', 'pitch', 'energy', 'pitch_cond'):
        assert maxdiff(pred[k].detach().cpu(), M['train/' + k]) < 5e-5, k
    worst, wk = 0.0, None
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        d = maxdiff(p.grad.cpu(), M['grad/' + k])
        if d > worst:
            worst, wk = d, k
    assert worst < 1e-4, (worst, wk)
    # generate (B=1) on the post-step weights captured by the reference
    m2 = _model(sub(M, 'gen_sd/'))
    out = m2.generate(torch.from_numpy(M['gen/x']).cuda(), torch.from_numpy(M['batch/speaker_emb'][:1]).cuda(), alpha=1.1)
    for k in ('mel', 'mel_post', 'dur', 'pitch', 'energy', 'pitch_cond'):
        assert out[k].shape == M['gen/' + k].shape, k
        assert maxdiff(out[k].float().cpu(), M['gen/' + k].astype(np.float32)) < 5e-5, k


def test_multispeaker_train_step_through_trainer():
    from forwardtacotron_amd.trainer import TrainStep
    M = load_npz('tiny_multi.npz')
    m = _model(sub(M, 'sd/'))
    ts = TrainStep(m, lr=float(M['lr']), train_cfg=TRAIN_CFG_MULTI)
    out = ts.step({k: v.clone().cuda() for k, v in sub(M, 'batch/').items()})
    assert abs(float(out['loss']) - float(M['loss/total'])) < 2e-5
    assert abs(float(out['grad_norm']) - float(M['grad_norm'])) < 1e-4 * max(1.0, float(M['grad_norm']))
    sd = m.state_dict()
    for k, v in sub(M, 'sd_after/').items():
        if v.dtype.is_floating_point:
            assert maxdiff(sd[k].cpu(), v) < 3e-5, k


MID_MULTI = dict(embed_dims=128, series_embed_dims=32, num_chars=135,
                 durpred_conv_dims=64, durpred_rnn_dims=32, durpred_dropout=0.0,
                 pitch_conv_dims=64, pitch_rnn_dims=32, pitch_dropout=0.0, pitch_strength=1.0,
                 pitch_cond_conv_dims=64, pitch_cond_rnn_dims=32, pitch_cond_dropout=0.0,
                 energy_conv_dims=64, energy_rnn_dims=32, energy_dropout=0.0, energy_strength=1.0,
                 rnn_dims=256, prenet_dims=256, prenet_k=4, postnet_num_highways=2,
                 prenet_dropout=0.0, postnet_dims=128, postnet_k=4, prenet_num_highways=2,
                 postnet_dropout=0.0, n_mels=80, speaker_emb_dims=256, pitch_cond_emb_dims=4,
                 pitch_cond_categorical_dims=3)


def _multi_batch(B, Tmax, seed):
    from oracle import ft_oracle as O
    batch = O.synthetic_batch(B=B, Tmax=Tmax, n_mels=80, seed=seed)
    batch['pitch_cond'] = ((batch['pitch'] != 0).long() + 1) * (batch['x'] > 0).long()
    se = torch.randn(B, 256, generator=torch.Generator().manual_seed(seed + 100))
    batch['speaker_emb'] = se / se.norm(dim=1, keepdim=True)
    return batch


def test_multispeaker_mid_size_train_step_vs_oracle():
    """one optimisation step of MultiForwardTacotron through TrainStep at sizes where the wide paths run
    -- 128x128 bf16-split GEMM tiles (forward, data and weight gradients), the 768-wide LSTM input of the shipped
    multispeaker model (2 x 256 prenet + 256 speaker), persistent recurrences incl. the reduce-scatter backward
    (G*H = 1024), one-launch bank weight gradients -- against oracle.multi_train_step: losses incl. the CrossEntropy
    term, gradient norm, every parameter after Adam, BN statistics (multi_forward_tacotron.py:186-241)."""
    from forwardtacotron_amd import hip as H
    from forwardtacotron_amd.multi_model import MultiForwardTacotron
    from forwardtacotron_amd.trainer import TrainStep
    from oracle import ft_oracle as O
    torch.manual_seed(31)
    m = MultiForwardTacotron(**MID_MULTI)
    assert m.lstm.weight_ih_l0.shape == (4 * 256, 768)
    P = {k: v.clone() for k, v in m.state_dict().items()}
    batch = _multi_batch(B=8, Tmax=96, seed=6)
    assert int(batch['mel_len'].max()) * 8 > 4000
    lr = 1e-3
    newP, _, info = O.multi_train_step(P, {}, {k: v.clone() for k, v in batch.items()}, MID_MULTI, TRAIN_CFG_MULTI, lr, 1)
    m = m.cuda()
    ts = TrainStep(m, lr=lr, train_cfg=TRAIN_CFG_MULTI)
    n0 = H.rnn_counters()[0]
    out = ts.step({k: v.clone().cuda() for k, v in batch.items()})
    ts.check()
    H.check_rnn_status()
    assert H.rnn_counters()[0] > n0                       # persistent recurrences really ran
    for k in ('loss', 'mel', 'mel_post', 'dur', 'pitch', 'energy', 'pitch_cond'):
        assert abs(float(out[k]) - float(info['losses'][k])) < 5e-5, k
    gn = float(info['grad_norm'])
    assert abs(float(out['grad_norm']) - gn) < 2e-4 * max(1.0, gn)
    # every gradient (TrainStep's flat gradient buffer still holds the raw, unclipped gradients after the step)
    gworst, gk = 0.0, None
    for k, p_ in m.named_parameters():
        d = maxdiff(p_.grad.cpu(), info['grads'][k])
        if d > gworst:
            gworst, gk = d, k
    assert gworst < 1e-4, (gworst, gk)
    sd = m.state_dict()
    worst, worst_k = 0.0, None
    for k, v in newP.items():
        if not v.dtype.is_floating_point:
            continue
        if 'running_' in k:
            assert maxdiff(sd[k].cpu(), v) < 1e-5 * max(1.0, float(v.abs().max())), k
        elif k in info['grads']:
            # Adam's first step moves a weight by lr * sign(g): compare where the gradient's sign is not rounding noise
            big = info['grads'][k].abs() > 1e-5
            if bool(big.any()):
                d = float((sd[k].cpu() - v).abs()[big].max())
                if d > worst:
                    worst, worst_k = d, k
    assert worst < 1e-4, (worst, worst_k)


def test_multispeaker_full_size_properties():
    """BASELINE configs[3] per GPU: multispeaker.yaml model, bs=64 (Tm = 811, 37,819 frames).  No oracle at this size:
    padding value reproduced exactly in the extra frame, eval deterministic, every output finite, and one TrainStep moves
    every parameter by at most lr (first Adam step) with a clean recurrence status."""
    from forwardtacotron_amd import data, hip as H
    from forwardtacotron_amd.multi_model import MultiForwardTacotron
    from forwardtacotron_amd.trainer import TrainStep
    torch.manual_seed(0)
    m = MultiForwardTacotron(**data.MULTISPEAKER_MODEL).cuda().eval()
    batch = {k: v.cuda() for k, v in _multi_batch(B=64, Tmax=128, seed=0).items()}
    Tm = int(batch['mel_len'].max())
    assert Tm == 811 and int(batch['mel_len'].sum()) == 37819
    dur0 = batch['dur'].clone()
    with torch.no_grad():
        a = m(batch)
        batch['dur'].copy_(dur0)
        b = m(batch)
    H.check_rnn_status()
    for k in a:
        assert torch.equal(a[k], b[k]), k
        assert bool(torch.isfinite(a[k]).all()), k
    assert tuple(a['mel'].shape) == (64, 80, Tm + 1) and tuple(a['pitch_cond'].shape) == (64, 128, 3)
    for k in ('mel', 'mel_post'):           # _pad: frame Tm (beyond every item) holds the padding value exactly
        assert bool((a[k][:, :, Tm:] == -11.5129).all()), k
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    ts = TrainStep(m, lr=1e-4, train_cfg=dict(data.SINGLESPEAKER_TRAIN, pitch_cond_loss_factor=0.1))
    batch['dur'].copy_(dur0)
    out = ts.step(batch)
    ts.check()
    H.check_rnn_status()
    assert bool(torch.isfinite(out['loss'])) and float(out['grad_norm']) > 0 and float(out['rnn_fault']) == 0.0
    worst = max(float((p.detach() - before[n]).abs().max()) for n, p in m.named_parameters())
    assert 0 < worst <= 1e-4 * 1.01
