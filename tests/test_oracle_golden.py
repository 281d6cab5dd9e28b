"""Pins oracle/ft_oracle.py against fixtures captured from the imported reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import ft_oracle as O
from helpers import TINY, TRAIN_CFG, load_npz, sub, maxdiff


@pytest.fixture(scope='module')
def L():
    return load_npz('layers.npz')


@pytest.fixture(scope='module')
def M():
    return load_npz('tiny_model.npz')


@pytest.mark.parametrize('tag,relu', [('bnc_k5', True), ('bnc_k4', True), ('bnc_k3_norelu', False),
                                      ('bnc_k1', True), ('bnc_k2', True)])
def test_batchnorm_conv(L, tag, relu):
    P = sub(L, tag + '/sd/')
    x = torch.from_numpy(L[tag + '/x'])
    y = O.batchnorm_conv(x, P, '', relu, False)
    assert maxdiff(y, L[tag + '/eval']) < 2e-6
    xg = x.clone().requires_grad_(True)
    Pg = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in P.items()}
    nb = {}
    y = O.batchnorm_conv(xg, Pg, '', relu, True, nb)
    assert maxdiff(y.detach(), L[tag + '/train']) < 5e-6
    (y * torch.from_numpy(L[tag + '/w'])).sum().backward()
    assert maxdiff(xg.grad, L[tag + '/dx']) < 2e-5
    assert maxdiff(Pg['conv.weight'].grad, L[tag + '/dW']) < 2e-5
    assert maxdiff(Pg['bnorm.weight'].grad, L[tag + '/dgamma']) < 2e-5
    assert maxdiff(Pg['bnorm.bias'].grad, L[tag + '/dbeta']) < 2e-5
    after = sub(L, tag + '/sd_after/')
    assert maxdiff(nb['bnorm.running_mean'], after['bnorm.running_mean']) < 1e-6
    assert maxdiff(nb['bnorm.running_var'], after['bnorm.running_var']) < 1e-6
    assert int(nb['bnorm.num_batches_tracked']) == int(after['bnorm.num_batches_tracked'])


def test_maxpool(L):
    assert maxdiff(O.maxpool_k2s1p1(torch.from_numpy(L['maxpool/x'])), L['maxpool/y']) == 0.0


def test_highway(L):
    y = O.highway(torch.from_numpy(L['highway/x']), sub(L, 'highway/sd/'), '')
    assert maxdiff(y, L['highway/y']) < 1e-6


def test_bigru(L):
    y = O.bigru(torch.from_numpy(L['gru/x']), sub(L, 'gru/sd/'), '')
    assert maxdiff(y, L['gru/y']) < 1e-6


def test_bilstm_packed_and_full(L):
    P = sub(L, 'lstm/sd/')
    x = torch.from_numpy(L['lstm/x'])
    lens = torch.from_numpy(L['lstm/lens'])
    y = O.bilstm(x, lens, P, '')
    assert maxdiff(y, L['lstm/y_packed']) < 1e-6
    assert maxdiff(O.bilstm(x, None, P, ''), L['lstm/y_full']) < 1e-6


def test_length_regulator(L):
    x = torch.from_numpy(L['lr/x'])
    dur = torch.from_numpy(L['lr/dur_in'].copy())
    y = O.length_regulate(x, dur)
    assert y.shape == L['lr/y'].shape
    assert np.array_equal(y.numpy(), L['lr/y'])                 # bit-exact copies
    assert np.array_equal(dur.numpy(), L['lr/dur_after'])       # in-place clamp
    r = O.lr_repeats(L['lr/dur_in'])
    assert r.tolist() == [[1, 1, 3, 0, 0, 1], [0, 0, 0, 1, 3, 0], [2] * 6]
    y0 = O.length_regulate(torch.zeros(2, 3, 4), torch.zeros(2, 3))
    assert list(y0.shape) == L['lr/zero_shape'].tolist()


def test_masked_l1(L):
    v = O.masked_l1(torch.from_numpy(L['l1/x']), torch.from_numpy(L['l1/t']), torch.from_numpy(L['l1/lens']))
    assert abs(float(v) - float(L['l1/loss'])) < 1e-6


def test_cbhg(L):
    P = sub(L, 'cbhg/sd/')
    x = torch.from_numpy(L['cbhg/x'])
    assert maxdiff(O.cbhg(x, P, '', 4, 2, False), L['cbhg/eval']) < 5e-6
    assert maxdiff(O.cbhg(x, P, '', 4, 2, True, {}), L['cbhg/train']) < 5e-6


def test_series_predictor(L):
    P = sub(L, 'sp/sd/')
    y = O.series_predictor(torch.from_numpy(L['sp/x']), P, '', False, alpha=2.0)
    assert maxdiff(y, L['sp/eval_alpha2']) < 2e-6


def test_pad(L):
    x = torch.from_numpy(L['pad/x'])
    assert np.array_equal(O.pad_to(x, 7).numpy(), L['pad/y7'])
    assert np.array_equal(O.pad_to(x, 4).numpy(), L['pad/y4'])


def test_tiny_model_eval_forward(M):
    P = sub(M, 'sd/')
    batch = sub(M, 'batch/')
    pred, _ = O.forward(P, batch, TINY, training=False)
    for k in ('mel', 'mel_post', 'dur', 'pitch', 'energy'):
        assert maxdiff(pred[k], M['eval/' + k]) < 2e-5, k


def test_tiny_model_train_step(M):
    P = sub(M, 'sd/')
    batch = sub(M, 'batch/')
    newP, opt, info = O.train_step(P, {}, batch, TINY, TRAIN_CFG, lr=float(M['lr']), step_count=1)
    for k in ('mel', 'mel_post', 'dur', 'pitch', 'energy'):
        assert maxdiff(info['pred'][k], M['train/' + k]) < 2e-5, k
    for k, gk in (('loss', 'total'), ('mel', 'mel'), ('mel_post', 'mel_post'), ('dur', 'dur'),
                  ('pitch', 'pitch'), ('energy', 'energy')):
        assert abs(float(info['losses'][k]) - float(M['loss/' + gk])) < 1e-5, k
    worst = 0.0
    for k, g in info['grads'].items():
        worst = max(worst, maxdiff(g, M['grad/' + k]))
    assert worst < 1e-4, worst
    assert abs(float(info['grad_norm']) - float(M['grad_norm'])) < 1e-4 * max(1.0, float(M['grad_norm']))
    after = sub(M, 'sd_after/')
    for k, v in after.items():
        if v.dtype.is_floating_point:
            # Adam's first step moves every weight by ~lr*sign(g): compare tightly
            assert maxdiff(newP[k], v) < 2e-5, k
        else:
            assert int(newP[k].reshape(-1)[0]) == int(v.reshape(-1)[0]), k


def test_generate():
    G = load_npz('generate.npz')
    P = sub(G, 'sd/')
    for tag in ('b1', 'b2'):
        out = O.generate(P, torch.from_numpy(G[tag + '/x']), TINY, alpha=0.9)
        for k in ('mel', 'mel_post', 'dur', 'pitch', 'energy'):
            assert out[k].shape == G[f'{tag}/{k}'].shape, (tag, k)
            assert maxdiff(out[k], G[f'{tag}/{k}']) < 2e-5, (tag, k)


def test_tiny_multispeaker_model():
    """MultiForwardTacotron restatement vs the imported reference (eval + full train step with CE loss)."""
    from helpers import TINY_MULTI, TRAIN_CFG_MULTI
    M = load_npz('tiny_multi.npz')
    P = sub(M, 'sd/')
    batch = sub(M, 'batch/')
    pred, _ = O.multi_forward(P, {k: v.clone() for k, v in batch.items()}, TINY_MULTI, training=False)
    for k in ('mel', 'mel_post', 'dur', 'pitch', 'energy', 'pitch_cond'):
        assert maxdiff(pred[k], M['eval/' + k]) < 2e-5, k
    newP, _, info = O.multi_train_step(P, {}, batch, TINY_MULTI, TRAIN_CFG_MULTI, float(M['lr']), 1)
    assert abs(float(info['losses']['loss']) - float(M['loss/total'])) < 1e-5
    assert abs(float(info['losses']['pitch_cond']) - float(M['loss/pitch_cond'])) < 1e-6
    worst = max(maxdiff(g, M['grad/' + k]) for k, g in info['grads'].items())
    assert worst < 1e-4, worst
    for k, v in sub(M, 'sd_after/').items():
        if v.dtype.is_floating_point:
            assert maxdiff(newP[k], v) < 2e-5, k


def test_expanded_length_beyond_mel_len_is_cut_at_the_packed_length():
    """tests/golden/make_golden_crop.py: T_lr = max(mel_len) + 5.  pad_packed_sequence returns max(mel_len) frames, so
    lin / postnet BatchNorm statistics / GRU see that many (forward_tacotron.py:147-152)."""
    M, C = load_npz('tiny_model.npz'), load_npz('tiny_crop.npz')
    P = sub(M, 'sd/')
    batch = sub(C, 'batch/')
    assert int(C['t_lr']) > int(batch['mel_len'].max())
    newP, _, info = O.train_step(P, {}, batch, TINY, TRAIN_CFG, lr=1e-3, step_count=1)
    for k in ('mel', 'mel_post', 'dur', 'pitch', 'energy'):
        assert maxdiff(info['pred'][k], C['train/' + k]) < 2e-5, k
    assert abs(float(info['losses']['loss']) - float(C['loss/total'])) < 1e-5
    assert max(maxdiff(g, C['grad/' + k]) for k, g in info['grads'].items()) < 1e-4
    for k, v in sub(C, 'sd_after/').items():
        assert maxdiff(info['new_buffers'][k] if 'new_buffers' in info else newP[k], v) < 1e-5, k
    # an item packed with more frames than its durations expand to: the reference's LSTM raises
    bad = sub(M, 'batch/')
    bad['mel_len'] = bad['mel_len'].clone()
    bad['mel_len'][int(bad['mel_len'].argmax())] += 2
    with pytest.raises(RuntimeError):
        O.forward(P, bad, TINY, training=True)
