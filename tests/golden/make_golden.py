"""Generate golden fixtures by IMPORTING the reference (only possible in the build
container, where /root/reference exists).  Output: small .npz files committed
under tests/golden/.  The reference's source never travels; these are data
(inputs + expected outputs) only.

    python tests/golden/make_golden.py

Fixtures:
  tiny_model.npz     tiny-config ForwardTacotron: state_dict, ragged 3-item batch,
                     eval outputs, train(dropout=0) outputs, 5 losses, all grads,
                     grad norm, post-Adam params, BN buffers after the step.
  layers.npz         per-layer known-answer vectors (BatchNormConv even/odd k,
                     maxpool shift, Highway, biGRU, packed biLSTM, LengthRegulator
                     rounding/negatives/zeros, _pad, MaskedL1, CBHG, SeriesPredictor).
  generate.npz       tiny-config generate() outputs (B=1 and B=2).
"""
import os
import sys

import numpy as np
import torch

REF = os.environ.get('FT_REFERENCE', '/root/reference')
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))

from models.forward_tacotron import ForwardTacotron, SeriesPredictor  # noqa: E402
from models.common_layers import (BatchNormConv, CBHG, HighwayNetwork,  # noqa: E402
                                  LengthRegulator)
from trainer.common import MaskedL1  # noqa: E402

TINY = dict(embed_dims=16, series_embed_dims=8, num_chars=135,
            durpred_conv_dims=16, durpred_rnn_dims=8, durpred_dropout=0.0,
            pitch_conv_dims=16, pitch_rnn_dims=12, pitch_dropout=0.0, pitch_strength=1.0,
            energy_conv_dims=16, energy_rnn_dims=8, energy_dropout=0.0, energy_strength=0.5,
            rnn_dims=20, prenet_dims=16, prenet_k=4, postnet_num_highways=2,
            prenet_dropout=0.0, postnet_dims=12, postnet_k=3, prenet_num_highways=2,
            postnet_dropout=0.0, n_mels=10)
TRAIN_CFG = dict(dur_loss_factor=0.1, pitch_loss_factor=0.1, energy_loss_factor=0.1,
                 clip_grad_norm=1.0)


def npd(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def randomize_bn(model, g):
    """Non-trivial BN affine + running stats so eval mode is a real test."""
    for name, mod in model.named_modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.weight.data = 1 + 0.2 * torch.randn(mod.weight.shape, generator=g)
            mod.bias.data = 0.1 * torch.randn(mod.bias.shape, generator=g)
            mod.running_mean.data = 0.1 * torch.randn(mod.running_mean.shape, generator=g)
            mod.running_var.data = 0.5 + torch.rand(mod.running_var.shape, generator=g)


def tiny_batch(g, n_mels):
    B, Tx = 3, 9
    x_len = torch.tensor([9, 5, 7])
    x = torch.zeros(B, Tx, dtype=torch.long)
    dur = torch.zeros(B, Tx)
    for b in range(B):
        L = int(x_len[b])
        x[b, :L] = torch.randint(1, 135, (L,), generator=g)
        dur[b, :L] = torch.randint(0, 5, (L,), generator=g).float()
    dur[0, 0] = 3.
    mel_len = dur.sum(1).long()
    Tm = int(mel_len.max())
    mel = torch.full((B, n_mels, Tm + 1), -11.5129)
    for b in range(B):
        n = int(mel_len[b])
        mel[b, :, :n] = torch.randn(n_mels, n, generator=g) * 2 - 5
    pitch = torch.randn(B, Tx, generator=g) * (x > 0)
    energy = torch.rand(B, Tx, generator=g) * (x > 0)
    return {'x': x, 'mel': mel, 'dur': dur, 'x_len': x_len, 'mel_len': mel_len,
            'pitch': pitch, 'energy': energy}


def make_tiny_model():
    torch.manual_seed(1234)
    g = torch.Generator().manual_seed(99)
    model = ForwardTacotron(**TINY)
    randomize_bn(model, g)
    batch = tiny_batch(g, TINY['n_mels'])
    out = {}
    for k, v in model.state_dict().items():
        out['sd/' + k] = v.clone().numpy()
    for k, v in batch.items():
        out['batch/' + k] = v.clone().numpy()

    # eval forward
    model.eval()
    with torch.no_grad():
        pred = model({k: v.clone() for k, v in batch.items()})
    for k, v in pred.items():
        out['eval/' + k] = v.numpy()

    # train forward + losses + grads + clip + Adam (forward_trainer.py:73-99)
    model.train()
    optim = torch.optim.Adam(model.parameters())
    lr = 1e-3
    for gr in optim.param_groups:
        gr['lr'] = lr
    b = {k: v.clone() for k, v in batch.items()}
    pitch_target = b['pitch'].detach().clone()
    energy_target = b['energy'].detach().clone()
    pred = model(b)
    l1 = MaskedL1()
    m1 = l1(pred['mel'], b['mel'], b['mel_len'])
    m2 = l1(pred['mel_post'], b['mel'], b['mel_len'])
    dl = l1(pred['dur'].unsqueeze(1), b['dur'].unsqueeze(1), b['x_len'])
    pl = l1(pred['pitch'], pitch_target.unsqueeze(1), b['x_len'])
    el = l1(pred['energy'], energy_target.unsqueeze(1), b['x_len'])
    loss = m1 + m2 + 0.1 * dl + 0.1 * pl + 0.1 * el
    optim.zero_grad()
    loss.backward()
    for k, v in pred.items():
        out['train/' + k] = v.detach().numpy()
    out['loss/total'] = loss.detach().numpy()
    out['loss/mel'] = m1.detach().numpy()
    out['loss/mel_post'] = m2.detach().numpy()
    out['loss/dur'] = dl.detach().numpy()
    out['loss/pitch'] = pl.detach().numpy()
    out['loss/energy'] = el.detach().numpy()
    for k, p in model.named_parameters():
        out['grad/' + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).clone().numpy()
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    out['grad_norm'] = gn.numpy()
    optim.step()
    for k, v in model.state_dict().items():
        out['sd_after/' + k] = v.clone().numpy()
    out['lr'] = np.float64(lr)
    np.savez_compressed(os.path.join(HERE, 'tiny_model.npz'), **out)
    print('tiny_model.npz', sum(v.nbytes for v in out.values()) // 1024, 'KiB raw')

    # generate
    gen = {}
    model.eval()
    g2 = torch.Generator().manual_seed(5)
    for tag, B in (('b1', 1), ('b2', 2)):
        x = torch.randint(1, 135, (B, 7), generator=g2)
        o = model.generate(x, alpha=0.9)
        gen[tag + '/x'] = x.numpy()
        for k, v in o.items():
            gen[f'{tag}/{k}'] = v.numpy()
    for k, v in model.state_dict().items():
        gen['sd/' + k] = v.clone().numpy()
    np.savez_compressed(os.path.join(HERE, 'generate.npz'), **gen)


def make_layers():
    g = torch.Generator().manual_seed(7)
    torch.manual_seed(77)
    out = {}

    # BatchNormConv: odd and even kernels, relu / no relu, train + eval
    for tag, (cin, cout, k, relu) in {'bnc_k5': (6, 8, 5, True), 'bnc_k4': (6, 8, 4, True),
                                      'bnc_k3_norelu': (5, 7, 3, False), 'bnc_k1': (4, 4, 1, True),
                                      'bnc_k2': (4, 6, 2, True)}.items():
        m = BatchNormConv(cin, cout, k, relu=relu)
        randomize_bn(m, g)
        x = torch.randn(3, cin, 11, generator=g)
        for kk, v in m.state_dict().items():
            out[f'{tag}/sd/{kk}'] = v.clone().numpy()
        out[f'{tag}/x'] = x.numpy()
        m.eval()
        out[f'{tag}/eval'] = m(x).detach().numpy()
        m.train()
        xg = x.clone().requires_grad_(True)
        y = m(xg)
        w = torch.randn(y.shape, generator=g)
        (y * w).sum().backward()
        out[f'{tag}/train'] = y.detach().numpy()
        out[f'{tag}/w'] = w.numpy()
        out[f'{tag}/dx'] = xg.grad.numpy()
        out[f'{tag}/dW'] = m.conv.weight.grad.numpy()
        out[f'{tag}/dgamma'] = m.bnorm.weight.grad.numpy()
        out[f'{tag}/dbeta'] = m.bnorm.bias.grad.numpy()
        for kk, v in m.state_dict().items():
            out[f'{tag}/sd_after/{kk}'] = v.clone().numpy()

    # maxpool shift
    mp = torch.nn.MaxPool1d(kernel_size=2, stride=1, padding=1)
    x = torch.randn(2, 3, 6, generator=g)
    out['maxpool/x'] = x.numpy()
    out['maxpool/y'] = mp(x)[:, :, :6].numpy()

    # Highway
    h = HighwayNetwork(6)
    h.W1.bias.data = 0.1 * torch.randn(6, generator=g)
    x = torch.randn(2, 5, 6, generator=g)
    for kk, v in h.state_dict().items():
        out[f'highway/sd/{kk}'] = v.clone().numpy()
    out['highway/x'] = x.numpy()
    out['highway/y'] = h(x).detach().numpy()

    # biGRU
    gru = torch.nn.GRU(5, 4, batch_first=True, bidirectional=True)
    x = torch.randn(3, 7, 5, generator=g)
    for kk, v in gru.state_dict().items():
        out[f'gru/sd/{kk}'] = v.clone().numpy()
    out['gru/x'] = x.numpy()
    out['gru/y'] = gru(x)[0].detach().numpy()

    # packed biLSTM with ragged lengths and -11.5129 padding
    from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
    lstm = torch.nn.LSTM(5, 6, batch_first=True, bidirectional=True)
    x = torch.randn(4, 9, 5, generator=g)
    lens = torch.tensor([4, 9, 1, 6])
    for kk, v in lstm.state_dict().items():
        out[f'lstm/sd/{kk}'] = v.clone().numpy()
    out['lstm/x'] = x.numpy()
    out['lstm/lens'] = lens.numpy()
    pk = pack_padded_sequence(x, lengths=lens, enforce_sorted=False, batch_first=True)
    y, _ = lstm(pk)
    y, _ = pad_packed_sequence(y, padding_value=-11.5129, batch_first=True)
    out['lstm/y_packed'] = y.detach().numpy()
    out['lstm/y_full'] = lstm(x)[0].detach().numpy()

    # LengthRegulator: rounding, negatives, zeros, ragged
    lr = LengthRegulator()
    x = torch.randn(3, 6, 4, generator=g)
    dur = torch.tensor([[1.49, 0.5, 2.5, 0., -3., 1.],
                        [0., 0., 0.49, 0.51, 3.2, 0.],
                        [2., 2., 2., 2., 2., 2.]])
    out['lr/x'] = x.numpy()
    out['lr/dur_in'] = dur.clone().numpy()
    y = lr(x, dur)
    out['lr/y'] = y.numpy()
    out['lr/dur_after'] = dur.numpy()          # mutated in place by the reference
    # all-zero durations -> empty time axis
    x0 = torch.randn(2, 3, 4, generator=g)
    out['lr/zero_shape'] = np.asarray(lr(x0, torch.zeros(2, 3)).shape)

    # MaskedL1
    l1 = MaskedL1()
    a = torch.randn(3, 5, 8, generator=g)
    t = torch.randn(3, 5, 8, generator=g)
    lens = torch.tensor([8, 3, 5])
    out['l1/x'] = a.numpy()
    out['l1/t'] = t.numpy()
    out['l1/lens'] = lens.numpy()
    out['l1/loss'] = l1(a, t, lens).numpy()

    # CBHG (K=4) train-mode fwd
    c = CBHG(K=4, in_channels=6, channels=8, proj_channels=[8, 6], num_highways=2, dropout=0.)
    randomize_bn(c, g)
    x = torch.randn(2, 6, 9, generator=g)
    for kk, v in c.state_dict().items():
        out[f'cbhg/sd/{kk}'] = v.clone().numpy()
    out['cbhg/x'] = x.numpy()
    c.eval()
    out['cbhg/eval'] = c(x).detach().numpy()
    c.train()
    out['cbhg/train'] = c(x).detach().numpy()

    # SeriesPredictor eval
    sp = SeriesPredictor(num_chars=20, emb_dim=4, conv_dims=6, rnn_dims=3, dropout=0.)
    randomize_bn(sp, g)
    xi = torch.randint(0, 20, (2, 7), generator=g)
    for kk, v in sp.state_dict().items():
        out[f'sp/sd/{kk}'] = v.clone().numpy()
    out['sp/x'] = xi.numpy()
    sp.eval()
    out['sp/eval_alpha2'] = sp(xi, alpha=2.0).detach().numpy()

    # _pad
    tm = ForwardTacotron(**TINY)
    x = torch.randn(2, 3, 5, generator=g)
    out['pad/x'] = x.numpy()
    out['pad/y7'] = tm._pad(x, 7).numpy()
    out['pad/y4'] = tm._pad(x, 4).numpy()

    np.savez_compressed(os.path.join(HERE, 'layers.npz'), **out)
    print('layers.npz', sum(v.nbytes for v in out.values()) // 1024, 'KiB raw')


TINY_MULTI = dict(embed_dims=16, series_embed_dims=8, num_chars=135,
                  durpred_conv_dims=16, durpred_rnn_dims=8, durpred_dropout=0.0,
                  pitch_conv_dims=16, pitch_rnn_dims=12, pitch_dropout=0.0, pitch_strength=1.0,
                  pitch_cond_conv_dims=12, pitch_cond_rnn_dims=8, pitch_cond_dropout=0.0,
                  energy_conv_dims=16, energy_rnn_dims=8, energy_dropout=0.0, energy_strength=0.5,
                  rnn_dims=20, prenet_dims=16, prenet_k=4, postnet_num_highways=2,
                  prenet_dropout=0.0, postnet_dims=12, postnet_k=3, prenet_num_highways=2,
                  postnet_dropout=0.0, n_mels=10, speaker_emb_dims=256, pitch_cond_emb_dims=4,
                  pitch_cond_categorical_dims=3)


def make_tiny_multi():
    """MultiForwardTacotron (models/multi_forward_tacotron.py) + the multi trainer's loss
    (trainer/multi_forward_trainer.py:73-99, restated here because the trainer module needs tensorboard)."""
    from models.multi_forward_tacotron import MultiForwardTacotron
    torch.manual_seed(4321)
    g = torch.Generator().manual_seed(77)
    model = MultiForwardTacotron(**TINY_MULTI)
    randomize_bn(model, g)
    batch = tiny_batch(g, TINY_MULTI['n_mels'])
    batch['pitch_cond'] = ((batch['pitch'] != 0).long() + 1) * (batch['x'] > 0).long()
    se = torch.randn(3, 256, generator=g)
    batch['speaker_emb'] = se / se.norm(dim=1, keepdim=True)
    out = {}
    for k, v in model.state_dict().items():
        out['sd/' + k] = v.clone().numpy()
    for k, v in batch.items():
        out['batch/' + k] = v.clone().numpy()
    model.eval()
    with torch.no_grad():
        pred = model({k: v.clone() for k, v in batch.items()})
    for k, v in pred.items():
        out['eval/' + k] = v.numpy()
    model.train()
    optim = torch.optim.Adam(model.parameters())
    lr = 1e-3
    for gr in optim.param_groups:
        gr['lr'] = lr
    b = {k: v.clone() for k, v in batch.items()}
    pitch_target = b['pitch'].detach().clone()
    energy_target = b['energy'].detach().clone()
    pred = model(b)
    l1 = MaskedL1()
    ce = torch.nn.CrossEntropyLoss(ignore_index=0)
    m1 = l1(pred['mel'], b['mel'], b['mel_len'])
    m2 = l1(pred['mel_post'], b['mel'], b['mel_len'])
    dl = l1(pred['dur'].unsqueeze(1), b['dur'].unsqueeze(1), b['x_len'])
    pl = l1(pred['pitch'], pitch_target.unsqueeze(1), b['x_len'])
    el = l1(pred['energy'], energy_target.unsqueeze(1), b['x_len'])
    pcl = ce(pred['pitch_cond'].transpose(1, 2), b['pitch_cond'])
    loss = m1 + m2 + 0.1 * dl + 0.1 * pl + 0.1 * el + 0.1 * pcl
    optim.zero_grad()
    loss.backward()
    for k, v in pred.items():
        out['train/' + k] = v.detach().numpy()
    out['loss/total'] = loss.detach().numpy()
    out['loss/pitch_cond'] = pcl.detach().numpy()
    for k, p in model.named_parameters():
        out['grad/' + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).clone().numpy()
    out['grad_norm'] = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0).numpy()
    optim.step()
    for k, v in model.state_dict().items():
        out['sd_after/' + k] = v.clone().numpy()
    out['lr'] = np.float64(lr)
    model.eval()
    x1 = torch.randint(1, 135, (1, 7), generator=g)
    gen = model.generate(x1, batch['speaker_emb'][:1], alpha=1.1)
    out['gen/x'] = x1.numpy()
    for k, v in gen.items():
        out['gen/' + k] = v.numpy()
    for k, v in model.state_dict().items():
        out['gen_sd/' + k] = v.clone().numpy()
    np.savez_compressed(os.path.join(HERE, 'tiny_multi.npz'), **out)
    print('tiny_multi.npz', sum(v.nbytes for v in out.values()) // 1024, 'KiB raw')


if __name__ == '__main__':
    which = sys.argv[1:] or ['model', 'layers', 'multi']
    if 'model' in which:
        make_tiny_model()
    if 'layers' in which:
        make_layers()
    if 'multi' in which:
        make_tiny_multi()
