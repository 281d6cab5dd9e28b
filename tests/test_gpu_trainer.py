"""GPU: full optimisation step (losses + backward + clip + Adam over flat buffers) vs the reference golden
and vs two consecutive oracle steps."""
import pytest
import torch

from helpers import TINY, TRAIN_CFG, load_npz, sub, maxdiff

pytestmark = pytest.mark.gpu


def _model(sd):
    from forwardtacotron_amd.model import ForwardTacotron
    m = ForwardTacotron(**TINY)
    m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    return m.cuda()


def test_train_step_matches_reference_golden():
    from forwardtacotron_amd.trainer import TrainStep
    M = load_npz('tiny_model.npz')
    m = _model(sub(M, 'sd/'))
    ts = TrainStep(m, lr=float(M['lr']), train_cfg=TRAIN_CFG)
    batch = {k: v.cuda() for k, v in sub(M, 'batch/').items()}
    out = ts.step(batch)
    assert abs(float(out['loss']) - float(M['loss/total'])) < 2e-5
    assert abs(float(out['grad_norm']) - float(M['grad_norm'])) < 1e-4 * max(1.0, float(M['grad_norm']))
    after = sub(M, 'sd_after/')
    sd = m.state_dict()
    assert list(sd.keys()) == list(after.keys())
    for k, v in after.items():
        if v.dtype.is_floating_point:
            assert maxdiff(sd[k].cpu(), v) < 3e-5, k
        else:
            assert int(sd[k].reshape(-1)[0]) == int(v.reshape(-1)[0]), k


def test_two_steps_match_oracle_and_flat_views_survive():
    from forwardtacotron_amd.trainer import TrainStep
    from oracle import ft_oracle as O
    M = load_npz('tiny_model.npz')
    P = sub(M, 'sd/')
    batch = sub(M, 'batch/')
    m = _model(P)
    ts = TrainStep(m, lr=2e-3, train_cfg=TRAIN_CFG)
    opt = {}
    for step in (1, 2):
        P, opt, info = O.train_step(P, opt, {k: v.clone() for k, v in batch.items()}, TINY, TRAIN_CFG, 2e-3, step)
        out = ts.step({k: v.clone().cuda() for k, v in batch.items()})
        assert abs(float(out['loss']) - float(info['losses']['loss'])) < 3e-5, step
    sd = m.state_dict()
    for k, v in P.items():
        if v.dtype.is_floating_point:
            assert maxdiff(sd[k].cpu(), v) < 1e-4, k
    assert ts.flat.attached()
    assert int(m.get_step()) == int(P['step'])


def test_dropout_is_seeded_and_scaled():
    from forwardtacotron_amd import hip as H
    x = torch.ones(1 << 16, device='cuda')
    a = H.dropout(x, 0.5, 123)
    b = H.dropout(x, 0.5, 123)
    c = H.dropout(x, 0.5, 124)
    assert torch.equal(a, b) and not torch.equal(a, c)
    keep = float((a > 0).float().mean())
    assert 0.48 < keep < 0.52
    assert set(a.unique().tolist()) == {0.0, 2.0}
