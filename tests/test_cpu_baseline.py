"""Pins oracle/ft_torch_cpu.py -- the stock-fused-torch-op CPU restatement that bench.py times as `cpu_baseline` --
against (a) the reference goldens (tests/golden/tiny_model.npz, tiny_crop.npz) and (b) the checker oracle
(oracle/ft_oracle.py) on a seeded batch with ragged lengths.  CPU only."""
import torch

from oracle import ft_oracle as O
from oracle import ft_torch_cpu as C
from helpers import TINY, ODD, TRAIN_CFG, load_npz, sub, maxdiff


def test_matches_reference_golden_train_step():
    M = load_npz('tiny_model.npz')
    tr = C.CpuTrainer(sub(M, 'sd/'), TINY, TRAIN_CFG, lr=float(M['lr']))
    info = tr.step(sub(M, 'batch/'))
    for k in ('mel', 'mel_post', 'dur', 'pitch', 'energy'):
        assert maxdiff(info['pred'][k], M['train/' + k]) < 1e-5, k
    assert abs(float(info['losses']['loss']) - float(M['loss/total'])) < 1e-5
    assert max(maxdiff(g, M['grad/' + k]) for k, g in info['grads'].items()) < 1e-4
    assert abs(float(info['grad_norm']) - float(M['grad_norm'])) < 1e-4 * max(1.0, float(M['grad_norm']))
    sd = tr.state_dict()
    for k, v in sub(M, 'sd_after/').items():
        if v.dtype.is_floating_point:
            assert maxdiff(sd[k], v) < 2e-5, k
        else:
            assert int(sd[k].reshape(-1)[0]) == int(v.reshape(-1)[0]), k


def test_matches_reference_golden_when_expansion_exceeds_mel_len():
    M, Cr = load_npz('tiny_model.npz'), load_npz('tiny_crop.npz')
    info = C.CpuTrainer(sub(M, 'sd/'), TINY, TRAIN_CFG, lr=1e-3).step(sub(Cr, 'batch/'))
    for k in ('mel', 'mel_post'):
        assert maxdiff(info['pred'][k], Cr['train/' + k]) < 1e-5, k
    assert max(maxdiff(g, Cr['grad/' + k]) for k, g in info['grads'].items()) < 1e-4


def test_two_steps_match_the_checker_oracle():
    from forwardtacotron_amd.model import ForwardTacotron
    torch.manual_seed(9)
    P = {k: v.clone() for k, v in ForwardTacotron(**ODD).state_dict().items()}
    batch = O.synthetic_batch(B=4, Tmax=11, n_mels=ODD['n_mels'], max_dur=5, seed=2)
    tr = C.CpuTrainer(P, ODD, TRAIN_CFG, lr=2e-3)
    opt = {}
    for step in (1, 2):
        P, opt, want = O.train_step(P, opt, {k: v.clone() for k, v in batch.items()}, ODD, TRAIN_CFG, 2e-3, step)
        got = tr.step(batch)
        assert abs(float(got['losses']['loss']) - float(want['losses']['loss'])) < 2e-5, step
        assert max(maxdiff(got['grads'][k], want['grads'][k]) for k in want['grads']) < 1e-4, step
    sd = tr.state_dict()
    for k, v in P.items():
        if v.dtype.is_floating_point:
            assert maxdiff(sd[k], v) < 1e-4, k
