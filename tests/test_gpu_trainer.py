"""GPU: full optimisation step (losses + backward + clip + Adam over flat buffers) vs the reference golden
and vs two consecutive oracle steps."""
import pytest
import torch

from helpers import TINY, TRAIN_CFG, load_npz, sub, maxdiff  # noqa: F401

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


# ---------------------------------------------------------------------------------------------------
# recurrence faults: a persistent recurrence whose poll runs out must never reach the parameters
# ---------------------------------------------------------------------------------------------------
def _force_timeouts(on: bool):
    from forwardtacotron_amd import _lib
    return _lib.lib().ft_rnn_set_max_spins(-1 if on else 0)


def test_recurrence_fault_is_sticky_across_later_launches():
    """a timed-out launch's status must survive later successful launches on the same workspace."""
    from forwardtacotron_amd import _lib, hip as H
    torch.manual_seed(0)
    B, T, Hh = 4, 40, 64
    xp = torch.randn(T, B, 6 * Hh, device='cuda')
    w = [torch.randn(3 * Hh, Hh, device='cuda') * 0.1 for _ in range(2)]
    b = [torch.randn(3 * Hh, device='cuda') * 0.1 for _ in range(2)]
    H.check_rnn_status()
    good, _ = H.gru_fwd(xp, w[0], w[1], b[0], b[1], Hh, save_gates=False)
    H.check_rnn_status()
    n0 = H.rnn_counters()[0]
    _force_timeouts(True)                     # fault injection: every arrival poll fails at once
    try:
        H.gru_fwd(xp, w[0], w[1], b[0], b[1], Hh, save_gates=False)
    finally:
        _force_timeouts(False)
    again, _ = H.gru_fwd(xp, w[0], w[1], b[0], b[1], Hh, save_gates=False)     # same workspace, succeeds
    assert H.rnn_counters()[0] == n0 + 2, 'the persistent form must have run (else this test tests nothing)'
    with pytest.raises(_lib.FtError):
        H.check_rnn_status(clear=False)       # ... and the earlier fault is still reported
    with pytest.raises(_lib.FtError):
        H.check_rnn_status()                  # reported again, now cleared
    H.check_rnn_status()
    assert torch.equal(good, again)


def _smoke_model():
    from forwardtacotron_amd.model import ForwardTacotron
    cfg = dict(TINY, rnn_dims=32, postnet_dims=32, durpred_rnn_dims=16)    # persistent-eligible recurrences (H % 16 == 0)
    torch.manual_seed(3)
    return ForwardTacotron(**cfg).cuda(), cfg


def test_recurrence_fault_skips_the_update_and_raises():
    """forced timeout -> parameters and Adam moments unchanged, grad_norm NaN, FtError surfaced by the
    NEXT step without anybody calling check_rnn_status; after that the trainer works again and matches a clean twin."""
    from forwardtacotron_amd import _lib, hip as H
    from forwardtacotron_amd.trainer import TrainStep
    from oracle import ft_oracle as O
    m, cfg = _smoke_model()
    twin, _ = _smoke_model()
    twin.load_state_dict(m.state_dict())
    batch = O.synthetic_batch(B=4, Tmax=12, n_mels=10, max_dur=5, seed=1)
    dev = lambda: {k: v.clone().cuda() for k, v in batch.items()}
    ts = TrainStep(m, lr=1e-3, train_cfg=TRAIN_CFG)
    ts2 = TrainStep(twin, lr=1e-3, train_cfg=TRAIN_CFG)
    H.check_rnn_status()
    ts.step(dev()); ts2.step(dev())
    ts.check()
    before = ts.flat.flat.clone()
    m1, v1 = ts.exp_avg.clone(), ts.exp_avg_sq.clone()
    n0 = H.rnn_counters()[0]
    _force_timeouts(True)
    try:
        out = ts.step(dev())
        torch.cuda.synchronize()
    finally:
        _force_timeouts(False)
    assert H.rnn_counters()[0] > n0
    assert float(out['rnn_fault']) == 1.0 and torch.isnan(out['grad_norm'])
    assert torch.equal(ts.flat.flat, before) and torch.equal(ts.exp_avg, m1) and torch.equal(ts.exp_avg_sq, v1)
    with pytest.raises(_lib.FtError, match='timed out'):
        ts.step(dev())                        # the previous step's flag has reached the host: raised BEFORE this step runs
    assert ts.opt_step == 1 and ts.skipped_steps == 1
    H.check_rnn_status()                      # the handler cleared the word
    out = ts.step(dev()); out2 = ts2.step(dev())
    ts.check(); ts2.check()
    assert float(out['rnn_fault']) == 0.0
    # the faulted step left no trace: parameters AND the forward-updated buffers (BatchNorm running statistics,
    # num_batches_tracked, step) agree with the twin that never ran it (they used to keep the faulted forward)
    assert maxdiff(ts.flat.flat.cpu(), ts2.flat.flat.cpu()) < 1e-6
    assert abs(float(out['grad_norm']) - float(out2['grad_norm'])) < 1e-5
    sd, sd2 = m.state_dict(), twin.state_dict()
    for k in sd:
        if sd[k].dtype.is_floating_point:
            assert bool(torch.isfinite(sd[k]).all()), k
            assert maxdiff(sd[k].cpu(), sd2[k].cpu()) < 1e-6, k
        else:
            assert torch.equal(sd[k].cpu(), sd2[k].cpu()), k


def test_recurrence_fault_fallback_policy_continues_on_per_step_kernels():
    from forwardtacotron_amd import _lib, hip as H
    from forwardtacotron_amd.trainer import TrainStep
    from oracle import ft_oracle as O
    m, cfg = _smoke_model()
    twin, _ = _smoke_model()
    twin.load_state_dict(m.state_dict())
    batch = O.synthetic_batch(B=4, Tmax=12, n_mels=10, max_dur=5, seed=1)
    dev = lambda: {k: v.clone().cuda() for k, v in batch.items()}
    ts = TrainStep(m, lr=1e-3, train_cfg=TRAIN_CFG, on_rnn_fault='fallback')
    H.check_rnn_status()
    _force_timeouts(True)
    try:
        ts.step(dev())
        torch.cuda.synchronize()
    finally:
        _force_timeouts(False)
    try:
        with pytest.warns(UserWarning, match='per-step'):
            ts.check()
        n0 = H.rnn_counters()[0]
        out = ts.step(dev())
        ts.check()
        assert H.rnn_counters()[0] == n0, 'per-step kernels expected after the fallback'
        assert float(out['rnn_fault']) == 0.0 and bool(torch.isfinite(out['grad_norm']))
        assert ts.opt_step == 1 and ts.skipped_steps == 1
        # forced fault + fallback: running statistics equal those of a clean twin that took the one real step on the
        # same (per-step) kernels, and are finite
        ts2 = TrainStep(twin, lr=1e-3, train_cfg=TRAIN_CFG)
        ts2.step(dev())
        ts2.check()
        sd, sd2 = m.state_dict(), twin.state_dict()
        for k in sd:
            if 'running_' in k:
                assert bool(torch.isfinite(sd[k]).all()), k
                assert maxdiff(sd[k].cpu(), sd2[k].cpu()) < 1e-6, k
            elif not sd[k].dtype.is_floating_point:
                assert torch.equal(sd[k].cpu(), sd2[k].cpu()), k
    finally:
        _lib.lib().ft_rnn_set_persistent(1)


@pytest.mark.parametrize('env', [{'FT_PRED_BWD_EARLY': '1'}, {'FT_STAGED_BACKWARD': '0'}, {'FT_PRED_STAGE_FIRST': '0'},
                                 {'FT_WGRAD_LATE': '0'}, {'FT_WGRAD_CUS': '0'}])
def test_step_schedule_variants_give_the_same_update(env, monkeypatch):
    """The backward of a step can be scheduled in several ways -- three stages with the predictors' stage first (the
    default where their BPTT grids fit beside the postnet GRU's) or in the middle, one backward call, the predictors' backward issued from inside the forward (model.predictor_hook), the
    weight-gradient stream unrestricted instead of limited to 28 CUs per XCD (ft_stream_create_cu_limited: fewer slots,
    so the weight-gradient planner picks other split counts -- the slabs are still summed in a fixed order, but a
    different one: this variant is compared to rounding, the others bit for bit) -- and every parameter's gradient
    is written by exactly one kernel launch in all of them: the update must not depend on the schedule, bit for bit.
    The batch has a negative duration: the LengthRegulator clamps it in place and the duration loss must see the clamped
    value whichever stream computes it first (forward_trainer.py:79-86)."""
    from forwardtacotron_amd.trainer import TrainStep
    M = load_npz('tiny_model.npz')
    batch = {k: v.clone() for k, v in sub(M, 'batch/').items()}
    lost = int(batch['dur'][0, 1])
    batch['dur'][0, 1] = -2.0
    batch['mel_len'][0] -= lost          # the item is packed with the frames its clamped durations still give

    def run():
        m = _model(sub(M, 'sd/'))
        ts = TrainStep(m, lr=2e-3, train_cfg=TRAIN_CFG)
        outs = [ts.step({k: v.clone().cuda() for k, v in batch.items()}) for _ in range(2)]
        torch.cuda.synchronize()
        return m.state_dict(), [{k: float(v) for k, v in o.items()} for o in outs]

    sd0, out0 = run()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    sd1, out1 = run()
    if 'FT_WGRAD_CUS' in env:               # other split counts: equal to rounding
        for a, b in zip(out0, out1):
            assert all(abs(a[k] - b[k]) <= 1e-5 * max(1.0, abs(a[k])) for k in a)
        for k in sd0:
            assert maxdiff(sd0[k].float(), sd1[k].float()) < 2e-6, k
        return
    assert out0 == out1
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]), k
