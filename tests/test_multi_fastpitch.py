"""MultiFastPitch: the oracle restatement is pinned against the fixture captured from the imported
reference (CPU), and the HIP model is checked against the same fixture through the C ABI (GPU)."""
import pytest
import torch

from helpers import TINY_MFP, TRAIN_CFG_MULTI, fp_state, load_npz, sub, maxdiff

OUT_KEYS = ('mel', 'mel_post', 'dur', 'pitch', 'energy', 'pitch_cond')


@pytest.fixture(scope='module')
def Z():
    return load_npz('tiny_multi_fastpitch.npz')


# ---------------------------------------------------------------------------------------------------
def test_oracle_eval_train_generate(Z):
    from oracle import fp_oracle as FP
    P = fp_state(Z, 'sd/')
    batch = sub(Z, 'batch/')
    with torch.no_grad():
        pred, _ = FP.multi_forward(P, batch, TINY_MFP, training=False)
    for k in OUT_KEYS:
        assert maxdiff(pred[k], Z['eval/' + k]) < 5e-6, k
    new_P, _, info = FP.multi_train_step(P, {}, batch, TINY_MFP, TRAIN_CFG_MULTI, float(Z['lr']), 1)
    assert maxdiff(info['losses']['loss'], Z['loss/total']) < 5e-6
    assert maxdiff(info['losses']['pitch_cond'], Z['loss/pitch_cond']) < 5e-6
    grads = sub(Z, 'grad/')
    assert set(grads) == set(info['grads'])
    for k, g in grads.items():
        assert maxdiff(info['grads'][k], g) < 2e-5 + 2e-5 * float(g.abs().max()), k
    assert maxdiff(info['grad_norm'], Z['grad_norm']) < 1e-4 * float(Z['grad_norm'])
    for k, v in fp_state(Z, 'sd_after/').items():
        if v.dtype.is_floating_point:
            live = grads[k].abs() > 1e-7 if k in grads else torch.ones_like(v, dtype=torch.bool)
            assert maxdiff(new_P[k][live], v[live]) < 2e-5, k
    G = fp_state(Z, 'gen_sd/')
    out = FP.multi_generate(G, torch.from_numpy(Z['gen/x']), torch.from_numpy(Z['batch/speaker_emb'][:1]), TINY_MFP,
                            alpha=1.1)
    for k in OUT_KEYS:
        assert out[k].shape == Z['gen/' + k].shape, k
        assert maxdiff(out[k].float(), Z['gen/' + k].astype('float32')) < 1e-5, k


# ---------------------------------------------------------------------------------------------------
def _model(sd):
    from forwardtacotron_amd.multi_fastpitch import MultiFastPitch
    m = MultiFastPitch(**TINY_MFP)
    m.load_state_dict(sd, strict=True)
    return m.cuda()


def test_state_dict_layout(Z):
    from forwardtacotron_amd.multi_fastpitch import MultiFastPitch
    ref_keys = [k[3:] for k in Z if k.startswith('sd/')]
    sd = MultiFastPitch(**TINY_MFP).state_dict()
    assert list(sd.keys()) == ref_keys
    for k in ref_keys:
        if not k.endswith('.pe'):
            assert tuple(sd[k].shape) == Z['sd/' + k].shape, k


@pytest.mark.gpu
def test_gpu_eval_train_generate(Z):
    from forwardtacotron_amd import ops
    m = _model(fp_state(Z, 'sd/'))
    batch = sub(Z, 'batch/')
    m.eval()
    with torch.no_grad():
        pred = m({k: v.clone().cuda() for k, v in batch.items()})
    for k in OUT_KEYS:
        assert pred[k].shape == Z['eval/' + k].shape, k
        assert maxdiff(pred[k].cpu(), Z['eval/' + k]) < 5e-5, k
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
    assert abs(float(ce) - float(Z['loss/pitch_cond'])) < 1e-5
    loss = loss + c['pitch_cond_loss_factor'] * ce
    assert abs(float(loss) - float(Z['loss/total'])) < 2e-5
    loss.backward()
    worst, wk = 0.0, None
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        d = maxdiff(p.grad.cpu(), Z['grad/' + k])
        if d > worst:
            worst, wk = d, k
    assert worst < 1e-4, (worst, wk)
    m2 = _model(fp_state(Z, 'gen_sd/'))
    out = m2.generate(torch.from_numpy(Z['gen/x']).cuda(), torch.from_numpy(Z['batch/speaker_emb'][:1]).cuda(),
                      alpha=1.1)
    for k in OUT_KEYS:
        assert out[k].shape == Z['gen/' + k].shape, k
        assert maxdiff(out[k].float().cpu(), Z['gen/' + k].astype('float32')) < 5e-5, k


@pytest.mark.gpu
def test_gpu_train_step_through_trainer(Z):
    from forwardtacotron_amd.trainer import TrainStep
    m = _model(fp_state(Z, 'sd/'))
    ts = TrainStep(m, lr=float(Z['lr']), train_cfg=TRAIN_CFG_MULTI)
    out = ts.step({k: v.clone().cuda() for k, v in sub(Z, 'batch/').items()})
    assert abs(float(out['loss']) - float(Z['loss/total'])) < 2e-5
    grads = sub(Z, 'grad/')
    # per-parameter gradients first: a wrong norm alone does not say WHICH gradient is off
    bad = []
    for n, p in m.named_parameters():           # p.grad is the parameter's view into the flat gradient buffer
        if n in grads:
            d = maxdiff(p.grad.cpu(), grads[n])
            if not d < 3e-5 + 3e-5 * float(abs(grads[n]).max()):
                bad.append((n, d, float(abs(grads[n]).max())))
    assert not bad, bad
    assert abs(float(out['grad_norm']) - float(Z['grad_norm'])) < 1e-4 * max(1.0, float(Z['grad_norm']))
    sd = m.state_dict()
    for k, v in sub(Z, 'sd_after/').items():
        if v.dtype.is_floating_point and not k.endswith('.pe'):
            live = grads[k].abs() > 1e-6
            assert maxdiff(sd[k].cpu()[live], v[live]) < 3e-5, k
