"""Golden index orders of the reference's BinnedLengthSampler (utils/dataset.py:54-83).

utils/dataset.py as a whole does not import here (librosa is absent), so this script compiles ONLY the
`BinnedLengthSampler` class node out of the reference's source (ast) in a namespace that holds what the class
uses (torch, numpy, random, Sampler) and runs it.  Output: tests/golden/sampler.npz -- for each case the lengths,
batch/bin sizes, the `random` seed and the index order the reference yields.  Data only; runs only in the build
container (needs /root/reference).

    python tests/golden/make_golden_sampler.py
"""
import ast
import os
import random

import numpy as np
import torch
from torch.utils.data import Sampler

REF = os.environ.get('FT_REFERENCE', '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [  # (n items, length range, batch_size, bin_size, data seed, random seed)
    (103, (10, 900), 4, 12, 0, 7),          # 8 full bins + remainder of 7
    (96, (10, 900), 32, 96, 1, 11),         # exactly one bin, no remainder
    (40, (50, 60), 8, 24, 2, 3),            # many equal lengths (sort ties), one bin + remainder
    (5, (1, 100), 2, 6, 3, 5),              # fewer items than one bin: remainder only is NOT reachable in the
                                            # reference (np.stack([]) raises) -> stored as raises=1
    (1000, (80, 1200), 32, 96, 4, 42),      # the shipped configuration's bin = 3 x batch
]


def reference_class():
    src = open(os.path.join(REF, 'utils', 'dataset.py')).read()
    node = next(n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == 'BinnedLengthSampler')
    ns = {'torch': torch, 'np': np, 'random': random, 'Sampler': Sampler}
    exec(compile(ast.Module(body=[node], type_ignores=[]), 'reference:utils/dataset.py', 'exec'), ns)
    return ns['BinnedLengthSampler']


def main():
    cls = reference_class()
    out = {'n_cases': np.asarray(len(CASES))}
    for c, (n, (lo, hi), bs, bin_size, dseed, rseed) in enumerate(CASES):
        lengths = np.random.RandomState(dseed).randint(lo, hi, size=n)
        out[f'c{c}_lengths'] = lengths
        out[f'c{c}_cfg'] = np.asarray([bs, bin_size, rseed])
        random.seed(rseed)
        try:
            sampler = cls(lengths.tolist(), bs, bin_size)
            out[f'c{c}_order'] = np.asarray([int(i) for i in sampler], dtype=np.int64)
            out[f'c{c}_raises'] = np.asarray(0)
            # second epoch of the SAME object (what a DataLoader does): the random stream continues, and the reference
            # shuffles views of its own index array, so the epoch starts from the previous epoch's within-bin order
            out[f'c{c}_order2'] = np.asarray([int(i) for i in sampler], dtype=np.int64)
        except ValueError:
            out[f'c{c}_raises'] = np.asarray(1)
    np.savez_compressed(os.path.join(HERE, 'sampler.npz'), **out)
    print('wrote sampler.npz', {k: v.shape for k, v in out.items() if k.endswith('order')})


if __name__ == '__main__':
    main()
