"""Checkpoint wire format + model dispatch (utils/checkpoints.py:13-49 of the reference)."""
import pytest
import torch

from helpers import TINY, TINY_FP, TINY_MFP, TINY_MULTI, TRAIN_CFG, load_npz, sub, maxdiff


def _cfg(kind, model_cfg):
    mc = {k: v for k, v in model_cfg.items() if k not in ('num_chars', 'n_mels')}
    return {'tts_model': kind, 'dsp': {'num_mels': model_cfg['n_mels']}, kind: {'model': mc}}


def test_init_tts_model_dispatch():
    from forwardtacotron_amd.checkpoints import init_tts_model
    from forwardtacotron_amd.fastpitch import FastPitch
    from forwardtacotron_amd.model import ForwardTacotron
    from forwardtacotron_amd.multi_fastpitch import MultiFastPitch
    from forwardtacotron_amd.multi_model import MultiForwardTacotron
    for kind, cfg, cls in (('forward_tacotron', TINY, ForwardTacotron), ('fast_pitch', TINY_FP, FastPitch),
                           ('multi_forward_tacotron', TINY_MULTI, MultiForwardTacotron),
                           ('multi_fast_pitch', TINY_MFP, MultiFastPitch)):
        config = _cfg(kind, cfg)
        m = init_tts_model(config)
        assert isinstance(m, cls)
        assert config[kind]['model']['num_chars'] == 135 and config[kind]['model']['n_mels'] == cfg['n_mels']  # mutated like the reference
        assert m.get_step() == 0
    assert isinstance(init_tts_model({'dsp': {'num_mels': 10}, 'forward_tacotron': {'model': {
        k: v for k, v in TINY.items() if k not in ('num_chars', 'n_mels')}}}), ForwardTacotron)   # default type
    with pytest.raises(ValueError):
        init_tts_model({'tts_model': 'tacotron9'})


@pytest.mark.gpu
def test_checkpoint_roundtrip_and_adam_wire_format(tmp_path):
    from forwardtacotron_amd.checkpoints import AdamStateAdapter, restore_checkpoint, save_checkpoint
    from forwardtacotron_amd.model import ForwardTacotron
    from forwardtacotron_amd.trainer import TrainStep
    M = load_npz('tiny_model.npz')
    sd0 = {k: torch.as_tensor(v) for k, v in sub(M, 'sd/').items()}
    batch = {k: v.cuda() for k, v in sub(M, 'batch/').items()}
    config = _cfg('forward_tacotron', TINY)

    def fresh():
        m = ForwardTacotron(**TINY)
        m.load_state_dict(sd0)
        m = m.cuda()
        return m, TrainStep(m, lr=float(M['lr']), train_cfg=TRAIN_CFG)

    def step(ts):
        return ts.step({k: v.clone() for k, v in batch.items()})

    m, ts = fresh()
    step(ts)
    # Adam moments after ONE step = (1-b1) g, (1-b2) g^2 of the clipped reference gradient
    coef = min(1.0, 1.0 / (float(M['grad_norm']) + 1e-6))
    osd = AdamStateAdapter(ts).state_dict()
    names = [n for n, _ in m.named_parameters()]
    assert osd['param_groups'][0]['params'] == list(range(len(names)))
    for i, n in enumerate(names):
        g = torch.as_tensor(M['grad/' + n]) * coef
        assert float(osd['state'][i]['step']) == 1.0
        assert maxdiff(osd['state'][i]['exp_avg'].cpu(), 0.1 * g) < 2e-6, n
        assert maxdiff(osd['state'][i]['exp_avg_sq'].cpu(), 0.001 * g * g) < 1e-7, n
    path = tmp_path / 'forward_step1.pt'
    save_checkpoint(m, ts, config, path, meta={'note': 'x'})
    ck = torch.load(path, map_location='cpu', weights_only=True)
    assert set(ck) == {'model', 'optim', 'config', 'note'}
    assert list(ck['model'].keys()) == [k[3:] for k in M if k.startswith('sd/')]        # the reference's 322 keys
    assert int(ck['model']['step']) == 1
    # the optimizer entry is a genuine torch.optim.Adam state_dict: stock Adam accepts it
    probe = ForwardTacotron(**TINY)
    stock = torch.optim.Adam(probe.parameters())
    stock.load_state_dict(ck['optim'])
    assert stock.param_groups[0]['lr'] == pytest.approx(float(M['lr']))
    assert maxdiff(stock.state[next(iter(probe.parameters()))]['exp_avg'], osd['state'][0]['exp_avg'].cpu()) == 0.0
    # ... and a stock Adam's state_dict loads into the fused step
    m_b, ts_b = fresh()
    AdamStateAdapter(ts_b).load_state_dict(stock.state_dict())
    assert ts_b.opt_step == 1 and maxdiff(ts_b.exp_avg.cpu(), ts.exp_avg.cpu()) == 0.0

    # resume == uninterrupted
    step(ts)
    m2, ts2 = fresh()
    restore_checkpoint(m2, ts2, path, torch.device('cuda'))
    assert m2.get_step() == 1 and ts2.opt_step == 1
    step(ts2)
    a, b = m.state_dict(), m2.state_dict()
    for k in a:
        assert torch.equal(a[k].cpu(), b[k].cpu()), k
    restore_checkpoint(m2, ts2, tmp_path / 'missing.pt', torch.device('cuda'))          # no file: no-op
