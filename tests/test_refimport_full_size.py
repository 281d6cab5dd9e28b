"""the CPU restatements against the LIVE import of the reference at the FULL model width
(configs/singlespeaker.yaml: 24.5 M parameters, 322 state_dict entries) -- the goldens under tests/golden pin them at
tiny widths only.  Needs /root/reference (build container); skipped on the GPU box, where the reference does not exist.
Batch kept small (B=3, 40 token slots, ~230 frames) so that the per-timestep checker oracle finishes in seconds; the
widths (bank K=16/8 x 256, highways, GRU-256/128, LSTM-512, predictors) are the shipped ones."""
import os
import sys

import pytest
import torch

from helpers import FULL, TRAIN_CFG, maxdiff

REF = os.environ.get('FT_REFERENCE', '/root/reference')
pytestmark = [pytest.mark.refimport,
              pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'models')), reason='reference not present')]


@pytest.fixture(scope='module')
def ref_step():
    """one hand-driven optimisation step of the imported reference (forward_trainer.py:73-99 with dropout 0)"""
    sys.path.insert(0, REF)
    try:
        from models.forward_tacotron import ForwardTacotron as RefModel
        from trainer.common import MaskedL1
    finally:
        sys.path.remove(REF)
    from oracle import ft_oracle as O
    cfg = dict(FULL, durpred_dropout=0.0, pitch_dropout=0.0, energy_dropout=0.0, prenet_dropout=0.0, postnet_dropout=0.0)
    torch.manual_seed(0)
    ref = RefModel(**cfg)
    P = {k: v.clone() for k, v in ref.state_dict().items()}
    batch = O.synthetic_batch(B=3, Tmax=40, n_mels=80, seed=11)
    b = {k: v.clone() for k, v in batch.items()}
    pt, et = b['pitch'].clone(), b['energy'].clone()
    ref.train()
    pred = ref(b)
    l1 = MaskedL1()
    loss = l1(pred['mel'], b['mel'], b['mel_len']) + l1(pred['mel_post'], b['mel'], b['mel_len']) \
        + 0.1 * l1(pred['dur'].unsqueeze(1), b['dur'].unsqueeze(1), b['x_len']) \
        + 0.1 * l1(pred['pitch'], pt.unsqueeze(1), b['x_len']) + 0.1 * l1(pred['energy'], et.unsqueeze(1), b['x_len'])
    loss.backward()
    grads = {k: p.grad.clone() for k, p in ref.named_parameters()}
    stats = {k: v.clone() for k, v in ref.state_dict().items() if 'running_' in k}
    return cfg, P, batch, {k: v.detach() for k, v in pred.items()}, float(loss.detach()), grads, stats


def test_state_dict_layout_of_the_drop_in_equals_the_import(ref_step):
    from forwardtacotron_amd.model import ForwardTacotron
    cfg, P = ref_step[0], ref_step[1]
    torch.manual_seed(0)
    sd = ForwardTacotron(**cfg).state_dict()
    assert list(sd.keys()) == list(P.keys()) and len(sd) == 322
    for k in sd:
        assert sd[k].shape == P[k].shape and sd[k].dtype == P[k].dtype, k
        assert torch.equal(sd[k], P[k]), k                       # identical default initialisation under the same seed


def test_stock_op_baseline_matches_the_import_at_full_width(ref_step):
    from oracle import ft_torch_cpu as C
    cfg, P, batch, pred, loss, grads, stats = ref_step
    info = C.CpuTrainer(P, cfg, TRAIN_CFG, lr=1e-3).step(batch)
    for k in ('mel', 'mel_post', 'dur', 'pitch', 'energy'):
        assert maxdiff(info['pred'][k], pred[k]) < 1e-5, k
    assert abs(float(info['losses']['loss']) - loss) < 1e-5
    assert max(maxdiff(info['grads'][k], g) for k, g in grads.items()) < 1e-4


def test_checker_oracle_matches_the_import_at_full_width(ref_step):
    from oracle import ft_oracle as O
    cfg, P, batch, pred, loss, grads, stats = ref_step
    newP, _, info = O.train_step(P, {}, {k: v.clone() for k, v in batch.items()}, cfg, TRAIN_CFG, 1e-3, 1)
    for k in ('mel', 'mel_post', 'dur', 'pitch', 'energy'):
        assert maxdiff(info['pred'][k], pred[k]) < 1e-5, k                # 1e-5 forward
    assert abs(float(info['losses']['loss']) - loss) < 1e-5
    worst = max(maxdiff(info['grads'][k], g) for k, g in grads.items())
    assert worst < 1e-4, worst                                             # 1e-4 gradients
    for k, v in stats.items():
        assert maxdiff(newP[k], v) < 1e-5, k
