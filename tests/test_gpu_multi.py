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
