"""Recurrent LAYERS restructured around their GEMMs.

(1) The input projection overlapped with the persistent recurrences (ft_*_layer_fwd: the input
projection arrives in time chunks behind gate words).  The chunked form must be BIT-identical to "projection whole, in
front": the same kernels form the same rows, only their launch order and streams differ.  (A recurrence that fills
whole XCDs -- the 512-wide LSTM -- stops every other dispatch while it is resident, profiles/r03_xcd_dispatch_probe.txt:
those shapes take the whole projection in front and only check that the call stays correct.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _weights(G, I, Hh, g):
    sh = 1.6 / Hh ** 0.5 if Hh > 128 else 0.3
    wih = [(torch.randn(G * Hh, I, generator=g) * 0.2).cuda() for _ in range(2)]
    whh = [(torch.randn(G * Hh, Hh, generator=g) * sh).cuda() for _ in range(2)]
    bih = [(torch.randn(G * Hh, generator=g) * 0.1).cuda() for _ in range(2)]
    bhh = [(torch.randn(G * Hh, generator=g) * 0.1).cuda() for _ in range(2)]
    return wih, whh, bih, bhh


def _same(ref, got, lens):
    """raw / cst are zero beyond an item's length; the saved activations are not written there at all"""
    T, B = ref[0].shape[:2]
    valid = torch.ones(T, B, dtype=torch.bool, device='cuda') if lens is None else \
        torch.arange(T, device='cuda')[:, None] < lens[None, :]
    for name, a, b in zip(('raw', 'cst', 'gates'), ref, got):
        assert torch.equal(a[valid], b[valid]), f'{name}: chunked projection changed the result'
    assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])


def _lstm_pair(B, T, I, Hh, lens, nchunks, lead=None, monkeypatch=None):
    from forwardtacotron_amd import hip as H
    g = torch.Generator().manual_seed(B + T + Hh)
    wih, whh, bih, bhh = _weights(4, I, Hh, g)
    x = torch.randn(B, T, I, generator=g).cuda()
    xp = H.linear_multi_fwd(x, wih, bih, y_tm_B=B)
    ref = H.lstm_fwd(xp, whh[0], whh[1], bhh[0], bhh[1], lens, Hh, True)
    if lead is not None:
        monkeypatch.setenv('FT_RNN_REV_LEAD', str(lead))
    got = H.lstm_layer_fwd(x, wih[0], wih[1], bih[0], bih[1], whh[0], whh[1], bhh[0], bhh[1], lens, Hh, True, nchunks)
    torch.cuda.synchronize()
    H.check_rnn_status()
    return ref, got


@pytest.mark.parametrize('B,T,I,Hh,packed,nchunks,lead', [
    (32, 841, 512, 512, True, 8, 4),          # the benchmark's decoder LSTM
    (32, 841, 512, 512, True, 8, 0),          # reverse direction waits for several chunks at its first step
    (32, 300, 96, 128, True, 16, 3),          # more chunks than the 16-step minimum allows at this T
    (19, 277, 64, 64, False, 5, 0),           # unpacked, ragged chunk at the end, B not a multiple of 16
    (7, 64, 32, 32, True, 4, 2),
])
def test_lstm_layer_forward_chunked_projection_is_bit_identical(B, T, I, Hh, packed, nchunks, lead, monkeypatch):
    lens = None
    if packed:
        g = torch.Generator().manual_seed(T)
        lens = torch.randint(max(1, T // 3), T + 1, (B,), generator=g)
        lens[0] = T
        lens[-1] = max(1, T // 3)
        lens = lens.cuda()
    ref, got = _lstm_pair(B, T, I, Hh, lens, nchunks, lead, monkeypatch)
    _same(ref, got, lens)


@pytest.mark.parametrize('B,T,I,Hh,nchunks', [(32, 841, 256, 256, 8), (32, 841, 256, 256, 3), (21, 130, 48, 64, 4)])
def test_gru_layer_forward_chunked_projection_is_bit_identical(B, T, I, Hh, nchunks):
    from forwardtacotron_amd import hip as H
    g = torch.Generator().manual_seed(B + T)
    wih, whh, bih, bhh = _weights(3, I, Hh, g)
    x = torch.randn(B, T, I, generator=g).cuda()
    xp = H.linear_multi_fwd(x, wih, bih, y_tm_B=B)
    ref = H.gru_fwd(xp, whh[0], whh[1], bhh[0], bhh[1], Hh, True)
    got = H.gru_layer_fwd(x, wih[0], wih[1], bih[0], bih[1], whh[0], whh[1], bhh[0], bhh[1], Hh, True, nchunks)
    torch.cuda.synchronize()
    H.check_rnn_status()
    assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])


def test_layer_forward_without_the_persistent_form_waits_for_every_chunk():
    """per-step kernels (persistent form off) read xp whole: the call must join the chunk stream first"""
    from forwardtacotron_amd import _lib
    old = _lib.lib().ft_rnn_set_persistent(0)
    try:
        lens = torch.tensor([40, 64, 17, 64, 33, 5, 64, 50], device='cuda')
        ref, got = _lstm_pair(8, 64, 32, 32, lens, 4)
    finally:
        _lib.lib().ft_rnn_set_persistent(old)
    _same(ref, got, lens)


def test_gate_poll_is_bounded_and_raises_the_sticky_fault():
    """fault injection (every poll fails at once): the gated launch drains, the fault word is set, nothing hangs"""
    from forwardtacotron_amd import _lib, hip as H
    g = torch.Generator().manual_seed(3)
    B, T, I, Hh = 16, 256, 32, 32
    wih, whh, bih, bhh = _weights(4, I, Hh, g)
    x = torch.randn(B, T, I, generator=g).cuda()
    H.check_rnn_status()
    _lib.lib().ft_rnn_set_max_spins(-1)
    try:
        H.lstm_layer_fwd(x, wih[0], wih[1], bih[0], bih[1], whh[0], whh[1], bhh[0], bhh[1], None, Hh, True, 8)
        torch.cuda.synchronize()
    finally:
        _lib.lib().ft_rnn_set_max_spins(0)
    with pytest.raises(Exception):
        H.check_rnn_status()
    H.check_rnn_status()            # cleared


# ---- LengthRegulator + decoder LSTM as one node: the input projection per token instead of per frame (ops.LRBiLSTMFn) ----
def _regulated_pair(B, Tx, I, Hh, packed, seed, zero_durs=False):
    from forwardtacotron_amd import model
    g = torch.Generator().manual_seed(seed)
    lstm = model.LSTM(I, Hh)
    with torch.no_grad():
        for p in lstm.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.2 if p.dim() == 1 else 1.0 / p.shape[1] ** 0.5))
    lstm = lstm.cuda()
    lr = model.LengthRegulator()
    x = torch.randn(B, Tx, I, generator=g)
    dur = torch.randint(1, 12, (B, Tx), generator=g).float()
    if zero_durs:
        dur[:, ::3] = 0.0           # tokens without frames
        dur[1, 5:] = 0.0            # an item much shorter than the rest
        dur[2, 0] = -3.0            # clamped in place, like the reference
    total = (dur.clamp(min=0) + 0.5).long().sum(1)
    lens = None
    if packed:
        lens = total.clone()
        lens[B // 2] = max(1, int(lens[B // 2]) - 3)      # packed shorter than the durations expand to
        lens = lens.cuda()
    Tm = int(lens.max()) if packed else int(total.max())
    w = torch.randn(B, Tm, 2 * Hh, generator=g).cuda()
    res = []
    for fused in (False, True):
        lstm.zero_grad()
        xg = x.cuda().requires_grad_(True)
        d = dur.clone().cuda()
        if fused:
            y = lstm.forward_regulated(xg, d, lens, -11.5129)
        else:
            y = lstm(lr(xg, d, lens), lens, -11.5129)
        (y * w).sum().backward()
        torch.cuda.synchronize()
        res.append((y.detach(), xg.grad, {k: v.grad.clone() for k, v in lstm.named_parameters()}, d))
    return res


@pytest.mark.parametrize('B,Tx,I,Hh,packed,zero_durs', [
    (32, 128, 512, 512, True, False),      # the benchmark's decoder: 4,096 token rows instead of ~26,900 frames
    (8, 33, 64, 64, True, True),
    (5, 17, 32, 32, False, True),          # generate path: padded length, frames beyond an item's end hold the bias
    (19, 40, 48, 128, True, False),
])
def test_regulated_lstm_equals_regulator_then_lstm(B, Tx, I, Hh, packed, zero_durs):
    from forwardtacotron_amd import hip as H
    a, b = _regulated_pair(B, Tx, I, Hh, packed, seed=B + Tx, zero_durs=zero_durs)
    H.check_rnn_status()
    assert torch.equal(a[3], b[3]), 'durations are clamped in place the same way'
    assert torch.equal(a[0], b[0]), 'forward: the per-token projection must give every frame the same bits'
    # backward: the same terms added per token first -- fp32-class differences
    ref = float(a[1].abs().max())
    assert float((a[1] - b[1]).abs().max()) <= 2e-5 * max(1.0, ref), 'd(input)'
    for k in a[2]:
        ref = float(a[2][k].abs().max())
        assert float((a[2][k] - b[2][k]).abs().max()) <= 2e-5 * max(1.0, ref), k


def test_xcd_fill_of_the_persistent_grids():
    """ft_rnn_xcd_fill_pct: the 512-wide LSTM holds every CU of its XCDs (nothing can be launched beside it -- the layer
    forward must not try to), the 256-wide GRU half of them; shapes that do not run persistent report -1"""
    from forwardtacotron_amd import _lib
    L = _lib.lib()
    assert L.ft_rnn_xcd_fill_pct(4, 0, 32, 841, 512) == 100
    assert L.ft_rnn_xcd_fill_pct(4, 1, 32, 841, 512) == 100
    assert L.ft_rnn_xcd_fill_pct(3, 0, 32, 841, 256) == 50
    assert L.ft_rnn_xcd_fill_pct(3, 1, 32, 841, 256) == 50
    assert L.ft_rnn_xcd_fill_pct(3, 1, 32, 128, 64) <= 25
    assert L.ft_rnn_xcd_fill_pct(3, 0, 4, 12, 12) == -1          # H % 16 != 0: per-step kernels
    assert 0 < L.ft_rnn_admit_budget_pct() <= 75
