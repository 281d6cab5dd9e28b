"""Data path feeding the train step: the reference's collators and length-binned sampler with the
same call signatures and outputs, plus what the MI355X step needs around them -- batches assembled once into PINNED
host memory and moved to HBM on a side stream one step ahead, so the step never waits on a host->device copy.

reference                                              here
utils/dataset.py:215-243  TacoCollator(r)            TacoCollator(r, pin_memory=False)
utils/dataset.py:246-270  ForwardCollator(taco)      ForwardCollator(taco)
utils/dataset.py:54-83    BinnedLengthSampler        BinnedLengthSampler (same `random`-driven order)
trainer/common.py:95-101  to_device                  DevicePrefetcher (async, double-buffered)  /  data.to_device

Padding rules (pinned by the reference's tests/test_collator.py, restated in tests/test_datapath.py):
x zero-padded int64 to max(x_len); mel padded with -11.5129 to max(mel_len)+1, rounded up to a multiple of r;
dur / pitch / energy zero-padded float32 and pitch_cond zero-padded int64 (each first cut to max(x_len));
x_len / mel_len int64; item_id / speaker_name python lists; speaker_emb stacked.
"""
import random
from typing import Any, Dict, Iterable, Iterator, List, Optional, Union

import numpy as np
import torch

MEL_PAD_VALUE = -11.5129


def _empty(shape, dtype: torch.dtype, pin: bool) -> torch.Tensor:
    t = torch.empty(shape, dtype=dtype)
    if pin and torch.cuda.is_available():
        t = t.pin_memory()
    return t


def _np_dtype_to_torch(a: np.ndarray) -> torch.dtype:
    return torch.from_numpy(np.empty(0, dtype=np.asarray(a).dtype)).dtype


class TacoCollator:
    """utils/dataset.py:215-243.  One output buffer per key, written in place (no per-item padded copies)."""

    def __init__(self, r: int, pin_memory: bool = False) -> None:
        self.r = r
        self.pin_memory = pin_memory

    def __call__(self, batch: List[Dict[str, Any]]) -> Dict[str, Any]:
        n = len(batch)
        x_len = torch.tensor([b['x_len'] for b in batch])
        max_x_len = int(max(x_len))
        text = _empty((n, max_x_len), torch.int64, self.pin_memory)
        text.zero_()
        for i, b in enumerate(batch):
            xi = np.asarray(b['x'])
            text[i, :len(xi)] = torch.from_numpy(xi.astype(np.int64, copy=False))
        spec_lens = [b['mel_len'] for b in batch]
        max_spec_len = max(spec_lens) + 1
        if max_spec_len % self.r != 0:
            max_spec_len += self.r - max_spec_len % self.r
        m0 = np.asarray(batch[0]['mel'])
        mel = _empty((n, m0.shape[0], max_spec_len), _np_dtype_to_torch(m0), self.pin_memory)
        mel.fill_(MEL_PAD_VALUE)
        for i, b in enumerate(batch):
            mi = np.asarray(b['mel'])
            mel[i, :, :mi.shape[-1]] = torch.from_numpy(mi)
        speaker_emb = torch.from_numpy(np.stack([np.asarray(b['speaker_emb']) for b in batch]))
        if self.pin_memory and torch.cuda.is_available():
            speaker_emb = speaker_emb.pin_memory()
        return {'x': text, 'mel': mel, 'item_id': [b['item_id'] for b in batch],
                'x_len': x_len, 'mel_len': torch.tensor(spec_lens),
                'speaker_emb': speaker_emb, 'speaker_name': [b['speaker_name'] for b in batch]}


class ForwardCollator:
    """utils/dataset.py:246-270"""

    def __init__(self, taco_collator: TacoCollator) -> None:
        self.taco_collator = taco_collator

    def _padded(self, batch, key: str, max_len: int, dtype: torch.dtype) -> torch.Tensor:
        out = _empty((len(batch), max_len), dtype, self.taco_collator.pin_memory)
        out.zero_()
        for i, b in enumerate(batch):
            v = np.asarray(b[key])[:max_len]
            out[i, :len(v)] = torch.from_numpy(np.ascontiguousarray(v)).to(dtype)
        return out

    def __call__(self, batch: List[Dict[str, Any]]) -> Dict[str, Any]:
        output = self.taco_collator(batch)
        max_x_len = int(max(b['x_len'] for b in batch))
        output.update({
            'pitch': self._padded(batch, 'pitch', max_x_len, torch.float32),
            'energy': self._padded(batch, 'energy', max_x_len, torch.float32),
            'dur': self._padded(batch, 'dur', max_x_len, torch.float32),
            'pitch_cond': self._padded(batch, 'pitch_cond', max_x_len, torch.int64),
        })
        return output


def _drawn_permutation(n: int) -> np.ndarray:
    """A permutation of range(n) drawn from python's global `random` with exactly the draws `random.shuffle` makes on
    a sequence of length n (the draws depend on the length only): shuffling x in place equals x[_drawn_permutation(n)]."""
    order = list(range(n))
    random.shuffle(order)
    return np.asarray(order, dtype=np.int64)


class BinnedLengthSampler(torch.utils.data.Sampler):
    """Length-binned epoch order with the reference's semantics (utils/dataset.py:54-83), written as permutations:

        ranks   = item indices in ascending length order                       (torch.sort, as the reference)
        table   = the first n_bins * bin_size ranks as a [n_bins, bin_size] table, tail = the rest
        epoch   = every table row permuted (row 0 first), then the ROWS permuted, then the tail permuted, concatenated

    Each permutation consumes python's global `random` stream exactly like one `random.shuffle` of that length, in
    that order, so the same `random.seed` gives the reference's epoch (tests/golden/sampler.npz holds orders captured
    from the reference class).  Like the reference -- which shuffles views of its own index array -- the within-row and
    tail permutations PERSIST in the object from one epoch to the next (the row permutation does not); a DataLoader
    that re-iterates one sampler therefore sees the reference's second epoch too.  Batches cut from an epoch are
    length-homogeneous, which keeps the padded GEMM shapes of a step tight.

    Deliberate difference: with fewer items than one bin the reference fails inside np.stack([]) (ValueError); here
    such a dataset is just its permuted tail."""

    def __init__(self, lengths, batch_size: int, bin_size: int):
        if bin_size % batch_size != 0:
            raise AssertionError('bin_size must be a multiple of batch_size')
        self.batch_size, self.bin_size = batch_size, bin_size
        ranks = torch.sort(torch.tensor(lengths).long())[1].numpy().copy()
        n_bins = len(ranks) // bin_size
        self._table = ranks[:n_bins * bin_size].reshape(n_bins, bin_size)
        self._tail = ranks[n_bins * bin_size:]

    def __iter__(self):
        table, tail = self._table, self._tail
        for r in range(table.shape[0]):
            table[r] = table[r][_drawn_permutation(table.shape[1])]
        parts = []
        if table.shape[0]:
            parts.append(table[_drawn_permutation(table.shape[0])].reshape(-1))
        if len(tail):
            tail[:] = tail[_drawn_permutation(len(tail))]
            parts.append(tail)
        epoch = np.concatenate(parts) if parts else np.empty(0, dtype=np.int64)
        return iter(torch.from_numpy(epoch.astype(np.int64, copy=True)))

    def __len__(self):
        return self._table.size + len(self._tail)


class DevicePrefetcher:
    """Wraps an iterable of collated host batches; yields batches whose tensors live on `device`.  The copy of batch
    i+1 is issued on a dedicated HIP stream while step i computes (pinned source buffers make it a true async DMA);
    the consumer's stream waits on the copy event, and the tensors are tagged with `record_stream` so the caching
    allocator does not recycle them under the consumer.  Non-tensor entries (item_id, speaker_name) pass through.
    With device='cpu' it degenerates to a plain iterator (used by the CPU tests)."""

    def __init__(self, loader: Iterable[Dict[str, Any]], device: Union[str, torch.device]):
        self.loader = loader
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == 'cuda' else None

    def _to_device(self, batch: Dict[str, Any]) -> Dict[str, Any]:
        if self.stream is None:
            return {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in batch.items()}
        out = {}
        with torch.cuda.stream(self.stream):
            for k, v in batch.items():
                out[k] = v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v
        return out

    def __iter__(self) -> Iterator[Dict[str, Any]]:
        it = iter(self.loader)
        nxt: Optional[Dict[str, Any]] = None
        try:
            nxt = self._to_device(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur = nxt
            if self.stream is not None:
                consumer = torch.cuda.current_stream(self.device)
                consumer.wait_stream(self.stream)
                for v in cur.values():
                    if torch.is_tensor(v):
                        v.record_stream(consumer)
            try:
                nxt = self._to_device(next(it))
            except StopIteration:
                nxt = None
            yield cur

    def __len__(self) -> int:
        return len(self.loader)  # type: ignore[arg-type]


def batches(items: List[Dict[str, Any]], lengths: List[int], batch_size: int, collator, bin_size: Optional[int] = None,
            drop_last: bool = False) -> Iterator[Dict[str, Any]]:
    """Minimal single-process loader (the reference trains with num_workers=0): length-binned order -> collated
    host batches."""
    sampler = BinnedLengthSampler(lengths, batch_size, bin_size or 3 * batch_size)
    order = [int(i) for i in sampler]
    for s in range(0, len(order), batch_size):
        chunk = order[s:s + batch_size]
        if drop_last and len(chunk) < batch_size:
            break
        yield collator([items[i] for i in chunk])
