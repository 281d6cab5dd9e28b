"""Pins oracle/fp_oracle.py (FastPitch restatement) against tests/golden/tiny_fastpitch.npz, captured from the
imported reference by tests/golden/make_golden_fastpitch.py.  CPU only."""
import pytest
import torch

from oracle import fp_oracle as FP
from helpers import TINY_FP, TRAIN_CFG, fp_state, load_npz, sub, maxdiff


@pytest.fixture(scope='module')
def Z():
    return load_npz('tiny_fastpitch.npz')


def test_eval_forward(Z):
    P = fp_state(Z, 'sd/')
    batch = sub(Z, 'batch/')
    with torch.no_grad():
        pred, _ = FP.forward(P, batch, TINY_FP, training=False)
    for k in ('mel', 'mel_post', 'dur', 'pitch', 'energy'):
        assert maxdiff(pred[k], Z['eval/' + k]) < 5e-6, k


def test_train_step(Z):
    P = fp_state(Z, 'sd/')
    batch = sub(Z, 'batch/')
    new_P, _, info = FP.train_step(P, {}, batch, TINY_FP, TRAIN_CFG, float(Z['lr']), 1)
    for k in ('mel', 'dur', 'pitch', 'energy'):
        assert maxdiff(info['pred'][k], Z['train/' + k]) < 5e-6, k
    for k, name in (('loss', 'total'), ('mel', 'mel'), ('mel_post', 'mel_post'), ('dur', 'dur'),
                    ('pitch', 'pitch'), ('energy', 'energy')):
        assert maxdiff(info['losses'][k], Z['loss/' + name]) < 5e-6, k
    grads = sub(Z, 'grad/')
    assert set(grads) == set(info['grads'])
    for k, g in grads.items():
        assert maxdiff(info['grads'][k], g) < 2e-5 + 2e-5 * float(g.abs().max()), k
    assert maxdiff(info['grad_norm'], Z['grad_norm']) < 1e-4 * float(Z['grad_norm'])
    after = fp_state(Z, 'sd_after/')
    assert set(after) == set(new_P)
    for k, v in after.items():
        if not v.dtype.is_floating_point:
            assert torch.equal(new_P[k], v), k
            continue
        # Adam's first step moves a weight by lr*sign(g): elements whose gradient is rounding noise only (e.g. the
        # key bias of every attention, whose true gradient is exactly 0) are excluded from the comparison
        live = grads[k].abs() > 1e-7 if k in grads else torch.ones_like(v, dtype=torch.bool)
        assert maxdiff(new_P[k][live], v[live]) < 2e-5, k


@pytest.mark.parametrize('tag,alpha', [('gen1', 0.9), ('gen2', 1.0)])
def test_generate(Z, tag, alpha):
    P = fp_state(Z, 'gen_sd/')
    x = torch.from_numpy(Z[tag + '/x'])
    out = FP.generate(P, x, TINY_FP, alpha=alpha)
    assert out['mel'].shape == Z[tag + '/mel'].shape
    for k in ('mel', 'mel_post', 'dur', 'pitch', 'energy'):
        assert maxdiff(out[k], Z[f'{tag}/{k}']) < 1e-5, k


def test_positional_encoding_refuses_sequences_beyond_the_pe_buffer():
    """FastPitch.generate with more than max_len = 5000 frames must raise (the reference fails with a shape
    error at common_layers.py:144), not index past the buffer.  Host-side check: runs without a GPU."""
    import pytest
    import torch
    from forwardtacotron_amd import _lib
    from forwardtacotron_amd.fastpitch import check_posenc_length
    pe = torch.zeros(5000, 1, 8)
    check_posenc_length(5000, pe)
    with pytest.raises(_lib.FtError, match='exceeds the pe buffer'):
        check_posenc_length(5001, pe)
