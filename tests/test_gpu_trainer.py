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


MID = dict(embed_dims=64, series_embed_dims=32, num_chars=135,
           durpred_conv_dims=64, durpred_rnn_dims=32, durpred_dropout=0.0,
           pitch_conv_dims=64, pitch_rnn_dims=32, pitch_dropout=0.0, pitch_strength=1.0,
           energy_conv_dims=64, energy_rnn_dims=32, energy_dropout=0.0, energy_strength=1.0,
           rnn_dims=256, prenet_dims=128, prenet_k=5, postnet_num_highways=2,
           prenet_dropout=0.0, postnet_dims=128, postnet_k=4, prenet_num_highways=2,
           postnet_dropout=0.0, n_mels=80)


def test_mid_size_train_step_vs_oracle():
    """One optimisation step through TrainStep at sizes where the wide paths run -- 128x128 bf16-split GEMM tiles
    (forward, data and weight gradients), conv-bank weight gradients from one launch (C % 128 == 0), persistent
    recurrences incl. the reduce-scatter backward (G*H = 1024), one-launch weight packs, token-side weight gradients on
    the main stream -- against the CPU oracle's step: losses, gradient norm, every parameter after Adam, BN statistics."""
    from forwardtacotron_amd import hip as H
    from forwardtacotron_amd.model import ForwardTacotron
    from forwardtacotron_amd.trainer import TrainStep
    from oracle import ft_oracle as O
    torch.manual_seed(21)
    m = ForwardTacotron(**MID)
    P = {k: v.clone() for k, v in m.state_dict().items()}
    batch = O.synthetic_batch(B=8, Tmax=96, n_mels=80, seed=4)
    assert int(batch['mel_len'].max()) * 8 > 4000          # enough rows for the wide tiles
    lr = 1e-3
    newP, _, info = O.train_step(P, {}, {k: v.clone() for k, v in batch.items()}, MID, TRAIN_CFG, lr, 1)
    m = m.cuda()
    ts = TrainStep(m, lr=lr, train_cfg=TRAIN_CFG)
    out = ts.step({k: v.clone().cuda() for k, v in batch.items()})
    H.check_rnn_status()
    assert abs(float(out['loss']) - float(info['losses']['loss'])) < 5e-5
    for k in ('mel', 'mel_post', 'dur', 'pitch', 'energy'):
        assert abs(float(out[k]) - float(info['losses'][k])) < 5e-5, k
    gn = float(info['grad_norm']) if 'grad_norm' in info else None
    if gn is not None:
        assert abs(float(out['grad_norm']) - gn) < 2e-4 * max(1.0, gn)
    sd = m.state_dict()
    worst, worst_k = 0.0, None
    for k, v in newP.items():
        if v.dtype.is_floating_point:
            # Adam's first step moves every weight by ~lr * sign(g): compare where the gradient is not rounding noise
            d = maxdiff(sd[k].cpu(), v)
            if 'running_' in k:
                assert d < 1e-5 * max(1.0, float(v.abs().max())), k
            elif k in info['grads']:
                big = info['grads'][k].abs() > 1e-6
                if bool(big.any()):
                    d = float((sd[k].cpu() - v).abs()[big].max())
                    if d > worst:
                        worst, worst_k = d, k
    assert worst < 1e-4, (worst, worst_k)


def test_deferred_weight_gradients_wait_for_the_emitting_stream():
    """A queued (GradSink.defer) weight gradient may be flushed from ANOTHER stream than the one that emitted it (a
    predictor's recurrence on its side stream flushes what the main stream queued): the side stream must then wait for
    the EMITTING stream's position, not the flusher's.  Stream A is held up by a long spin before it produces the
    operand; the flush comes from an idle stream B right away."""
    from forwardtacotron_amd import ops
    dev = torch.device('cuda')
    w = torch.zeros(1 << 20, device=dev)
    view = torch.full_like(w, -1.0)
    src = torch.zeros_like(w)
    A, B_, side = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    sink = ops.GradSink({w.data_ptr(): (0, view)}, stream=side)
    sink.defer = True
    sink.begin_step()
    ops.set_grad_sink(sink)
    try:
        torch.cuda.synchronize()
        with torch.cuda.stream(A):
            torch.cuda._sleep(200_000_000)                  # ~0.1 s: the operand is NOT ready for a long time
            src.fill_(3.0)
            r = ops._emit(w, lambda out: out.copy_(src), (src.view(1024, -1),))
            assert r is None and sink.pending and 0 in sink.held
        with torch.cuda.stream(B_):
            ops.flush_deferred()
        assert not sink.pending and not sink.held
        torch.cuda.synchronize()
        assert float(view.min()) == 3.0 and float(view.max()) == 3.0
    finally:
        ops.set_grad_sink(None)
