"""Data path.  The collator checks restate the reference's own tests/test_collator.py:40-65 (same
items, same expected values); the sampler is checked against index
orders captured from the reference's class (tests/golden/sampler.npz, made by tests/golden/make_golden_sampler.py) and
through the properties its algorithm guarantees (utils/dataset.py:54-83)."""
import os
import random

import numpy as np
import pytest
import torch

from forwardtacotron_amd.datapath import (BinnedLengthSampler, DevicePrefetcher, ForwardCollator, TacoCollator,
                                          batches)


def _items():
    return [
        {'item_id': 0, 'mel': np.full((2, 5), fill_value=1.), 'x': np.full(2, fill_value=2.), 'mel_len': 5, 'x_len': 2,
         'dur': np.full(2, fill_value=3.), 'pitch': np.full(2, fill_value=4.), 'pitch_cond': np.full(2, fill_value=5.),
         'energy': np.full(2, fill_value=5.), 'speaker_emb': np.full(1, fill_value=4.), 'speaker_name': 'speaker_1'},
        {'item_id': 1, 'mel': np.full((2, 6), fill_value=1.), 'x': np.full(3, fill_value=2.), 'mel_len': 6, 'x_len': 3,
         'dur': np.full(3, fill_value=3.), 'pitch': np.full(3, fill_value=4.), 'pitch_cond': np.full(3, fill_value=5.),
         'energy': np.full(3, fill_value=5.), 'speaker_emb': np.full(1, fill_value=5.), 'speaker_name': 'speaker_2'},
    ]


def test_collate_forward_matches_reference_test():
    batch = ForwardCollator(taco_collator=TacoCollator(r=1))(_items())
    assert batch['item_id'] == [0, 1]
    assert tuple(batch['mel'][0].size()) == (2, 7) and tuple(batch['mel'][1].size()) == (2, 7)
    assert torch.sum(batch['mel'][0], dim=0).tolist() == [2., 2., 2., 2., 2., -11.5129 * 2, -11.5129 * 2]
    assert torch.sum(batch['mel'][1], dim=0).tolist() == [2., 2., 2., 2., 2., 2., -11.5129 * 2]
    assert batch['x_len'].tolist() == [2, 3] and batch['mel_len'].tolist() == [5, 6]
    assert batch['x'][0].tolist() == [2., 2., 0] and batch['x'][1].tolist() == [2., 2., 2.]
    assert batch['dur'][0].tolist() == [3., 3., 0] and batch['dur'][1].tolist() == [3., 3., 3.]
    assert batch['pitch'][0].tolist() == [4., 4., 0] and batch['pitch'][1].tolist() == [4., 4., 4.]
    assert batch['pitch_cond'][0].tolist() == [5., 5., 0.] and batch['pitch_cond'][1].tolist() == [5., 5., 5.]
    assert batch['energy'][0].tolist() == [5., 5., 0] and batch['energy'][1].tolist() == [5., 5., 5.]
    assert batch['speaker_emb'][0].tolist() == [4.] and batch['speaker_emb'][1].tolist() == [5.]
    assert batch['speaker_name'] == ['speaker_1', 'speaker_2']
    # dtypes of the reference's outputs (utils/dataset.py:223,252-259)
    assert batch['x'].dtype == torch.int64 and batch['pitch_cond'].dtype == torch.int64
    assert batch['dur'].dtype == batch['pitch'].dtype == batch['energy'].dtype == torch.float32
    assert batch['x_len'].dtype == batch['mel_len'].dtype == torch.int64


def test_collate_reduction_factor_and_cut():
    items = _items()
    items[0]['dur'] = np.full(5, fill_value=3.)          # longer than max(x_len): cut first (dataset.py:252)
    batch = ForwardCollator(TacoCollator(r=4))(items)
    assert batch['mel'].shape[-1] == 8                   # max(mel_len)+1 = 7 -> next multiple of r
    assert float(batch['mel'][0, 0, 5]) == pytest.approx(-11.5129)
    assert batch['dur'][0].tolist() == [3., 3., 3.]
    f32 = [dict(it, mel=it['mel'].astype(np.float32)) for it in _items()]
    assert TacoCollator(r=1)(f32)['mel'].dtype == torch.float32


GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'sampler.npz')


def test_binned_length_sampler_matches_reference_orders():
    """orders captured from the reference class by tests/golden/make_golden_sampler.py (incl. the second epoch of
    one sampler object, whose start state the reference's in-place shuffles leave behind)"""
    g = np.load(GOLD)
    seen = 0
    for c in range(int(g['n_cases'])):
        lengths = g[f'c{c}_lengths'].tolist()
        bs, bin_size, rseed = (int(v) for v in g[f'c{c}_cfg'])
        s = BinnedLengthSampler(lengths, batch_size=bs, bin_size=bin_size)
        assert len(s) == len(lengths)
        random.seed(rseed)
        if int(g[f'c{c}_raises']):
            # fewer items than a bin: the reference dies in np.stack([]); here the epoch is the permuted tail
            assert sorted(int(i) for i in s) == list(range(len(lengths)))
            continue
        assert [int(i) for i in s] == g[f'c{c}_order'].tolist()
        assert [int(i) for i in s] == g[f'c{c}_order2'].tolist()
        seen += 1
    assert seen >= 4


def test_binned_length_sampler_properties():
    g = np.random.RandomState(0)
    lengths = g.randint(10, 900, size=103).tolist()
    s = BinnedLengthSampler(lengths, batch_size=4, bin_size=12)
    random.seed(7)
    order = [int(i) for i in s]
    assert len(s) == 103 and sorted(order) == list(range(103))           # a permutation
    rank = {int(i): r for r, i in enumerate(torch.sort(torch.tensor(lengths).long())[1].tolist())}
    # every full bin is a contiguous range of the length-sorted order; the remainder (7 longest) comes last
    for b in range(103 // 12):
        ranks = sorted(rank[i] for i in order[b * 12:(b + 1) * 12])
        assert ranks == list(range(ranks[0], ranks[0] + 12)) and ranks[0] % 12 == 0
    assert sorted(rank[i] for i in order[96:]) == list(range(96, 103))
    with pytest.raises(AssertionError):
        BinnedLengthSampler(lengths, batch_size=5, bin_size=12)
    assert [int(i) for i in BinnedLengthSampler([], 2, 4)] == []


def _dataset(n=20, seed=3):
    g = np.random.RandomState(seed)
    items = []
    for i in range(n):
        L = int(g.randint(3, 9))
        dur = g.randint(1, 5, size=L)
        items.append({'item_id': f'id{i}', 'x': g.randint(1, 135, size=L), 'x_len': L,
                      'mel': g.randn(6, int(dur.sum())).astype(np.float32), 'mel_len': int(dur.sum()),
                      'dur': dur.astype(np.float32), 'pitch': g.randn(L).astype(np.float32),
                      'energy': g.rand(L).astype(np.float32), 'pitch_cond': g.randint(1, 3, size=L),
                      'speaker_emb': g.randn(4).astype(np.float32), 'speaker_name': 'spk'})
    return items


def test_loader_and_prefetcher_cpu():
    items = _dataset()
    random.seed(0)
    host = list(batches(items, [it['mel_len'] for it in items], 4, ForwardCollator(TacoCollator(1)), bin_size=8))
    assert sum(len(b['item_id']) for b in host) == 20
    random.seed(0)
    pre = list(DevicePrefetcher(batches(items, [it['mel_len'] for it in items], 4, ForwardCollator(TacoCollator(1)),
                                        bin_size=8), 'cpu'))
    assert len(pre) == len(host)
    for a, b in zip(host, pre):
        assert a['item_id'] == b['item_id']
        for k in ('x', 'mel', 'dur', 'pitch', 'energy', 'pitch_cond', 'x_len', 'mel_len', 'speaker_emb'):
            assert torch.equal(a[k], b[k]), k
        assert int(b['mel'].shape[-1]) == int(b['mel_len'].max()) + 1
        for i in range(len(b['item_id'])):                                  # sum(dur) == mel_len, pad value beyond
            n = int(b['mel_len'][i])
            assert float(b['dur'][i].sum()) == n
            assert bool((b['mel'][i, :, n:] == torch.tensor(-11.5129)).all())


@pytest.mark.gpu
def test_gta_feature_dump(tmp_path):
    """train_forward.py:33-51: one <item_id>.npy per item = mel_post[:, :mel_len] of the eval forward."""
    from forwardtacotron_amd.gta import create_gta_features
    from forwardtacotron_amd.model import ForwardTacotron
    from helpers import TINY
    items = _dataset(n=10, seed=9)
    for it in items:
        it['mel'] = np.random.RandomState(it['mel_len']).randn(10, it['mel_len']).astype(np.float32)
    coll = ForwardCollator(TacoCollator(1, pin_memory=True))
    random.seed(2)
    host = list(batches(items, [it['mel_len'] for it in items], 4, coll, bin_size=4))
    torch.manual_seed(0)
    model = ForwardTacotron(**TINY).cuda()
    n = create_gta_features(model, host[:2], host[2:], tmp_path)
    assert n == 10 and not model.training
    for hb in host:
        with torch.no_grad():
            want = model({k: (v.cuda() if torch.is_tensor(v) else v) for k, v in hb.items()})['mel_post'].cpu().numpy()
        for j, item_id in enumerate(hb['item_id']):
            got = np.load(tmp_path / f'{item_id}.npy', allow_pickle=False)
            assert got.shape == (10, int(hb['mel_len'][j]))
            assert np.array_equal(got, want[j][:, :int(hb['mel_len'][j])])


@pytest.mark.gpu
def test_prefetcher_gpu_pinned_async_and_feeds_the_step():
    from forwardtacotron_amd.model import ForwardTacotron
    from helpers import TINY
    items = _dataset(n=12, seed=5)
    for it in items:                                        # TINY model: 10 mel channels
        it['mel'] = np.random.RandomState(it['x_len']).randn(10, it['mel_len']).astype(np.float32)
    coll = ForwardCollator(TacoCollator(1, pin_memory=True))
    random.seed(1)
    host = list(batches(items, [it['mel_len'] for it in items], 4, coll, bin_size=4))
    assert all(b['mel'].is_pinned() and b['x'].is_pinned() and b['dur'].is_pinned() for b in host)
    model = ForwardTacotron(**TINY).cuda().eval()
    n = 0
    for hb, db in zip(host, DevicePrefetcher(host, 'cuda')):
        for k in ('x', 'mel', 'dur', 'pitch', 'energy', 'x_len', 'mel_len'):
            assert db[k].is_cuda and torch.equal(db[k].cpu(), hb[k]), k
        assert db['item_id'] == hb['item_id']
        with torch.no_grad():
            out = model(db)
        assert out['mel'].shape == db['mel'].shape
        n += 1
    assert n == len(host)
