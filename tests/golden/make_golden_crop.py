"""Golden for the case the other fixtures do not reach: the rounded durations of an item expand to MORE frames than
its mel_len, so T_lr > max(mel_len).  The reference's pad_packed_sequence (forward_tacotron.py:147-152) then returns
max(mel_len) frames -- `lin`, the postnet's BatchNorm statistics and its GRU see that many, not T_lr.  Imports the
reference (build container only); reuses the tiny model of tiny_model.npz.  Output: tiny_crop.npz (batch, train-mode
outputs, losses, every gradient, BN buffers after the forward).

    python tests/golden/make_golden_crop.py
"""
import os
import sys

import numpy as np
import torch

REF = os.environ.get('FT_REFERENCE', '/root/reference')
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))

from models.forward_tacotron import ForwardTacotron  # noqa: E402
from trainer.common import MaskedL1  # noqa: E402

sys.path.insert(0, os.path.dirname(HERE))
from helpers import TINY, load_npz, sub  # noqa: E402


def main():
    M = load_npz('tiny_model.npz')
    model = ForwardTacotron(**TINY)
    model.load_state_dict(sub(M, 'sd/'))
    batch = {k: v.clone() for k, v in sub(M, 'batch/').items()}
    # the LONGEST item gets 4 more frames of duration than its mel_len (and one fractional duration that rounds up):
    # T_lr = max(mel_len) + 5 > max(mel_len)
    b = int(batch['mel_len'].argmax())
    batch['dur'][b, 1] += 4.0
    batch['dur'][b, 2] += 0.6
    r = (batch['dur'].clamp(min=0) + 0.5).long().sum(1)
    assert int(r.max()) > int(batch['mel_len'].max()) and bool((r >= batch['mel_len']).all())
    out = {'batch/' + k: v.clone().numpy() for k, v in batch.items()}
    model.train()
    bb = {k: v.clone() for k, v in batch.items()}
    pt, et = bb['pitch'].clone(), bb['energy'].clone()
    pred = model(bb)
    l1 = MaskedL1()
    m1 = l1(pred['mel'], bb['mel'], bb['mel_len'])
    m2 = l1(pred['mel_post'], bb['mel'], bb['mel_len'])
    dl = l1(pred['dur'].unsqueeze(1), bb['dur'].unsqueeze(1), bb['x_len'])
    pl = l1(pred['pitch'], pt.unsqueeze(1), bb['x_len'])
    el = l1(pred['energy'], et.unsqueeze(1), bb['x_len'])
    loss = m1 + m2 + 0.1 * dl + 0.1 * pl + 0.1 * el
    loss.backward()
    for k, v in pred.items():
        out['train/' + k] = v.detach().numpy()
    out['loss/total'] = loss.detach().numpy()
    for k, p in model.named_parameters():
        out['grad/' + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).clone().numpy()
    for k, v in model.state_dict().items():
        if 'running_' in k:
            out['sd_after/' + k] = v.clone().numpy()
    out['t_lr'] = np.asarray(int(r.max()))
    np.savez_compressed(os.path.join(HERE, 'tiny_crop.npz'), **out)
    print('tiny_crop.npz: T_lr', int(r.max()), 'max mel_len', int(batch['mel_len'].max()))

    # and the error case: an item packed with more frames than its durations expand to
    bad = {k: v.clone() for k, v in sub(M, 'batch/').items()}
    bad['mel_len'] = bad['mel_len'].clone()
    bad['mel_len'][b] += 2
    bad['mel'] = torch.nn.functional.pad(bad['mel'], [0, 2], value=-11.5129)
    try:
        model(bad)
        print('reference did NOT raise')
    except RuntimeError as e:
        print('reference raises:', str(e)[:80])


if __name__ == '__main__':
    main()
