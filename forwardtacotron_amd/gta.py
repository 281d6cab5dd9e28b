"""Ground-truth-aligned feature dump (train_forward.py:33-51 of the reference): run the trained model in eval mode
over the train + validation batches and save each item's `mel_post[:, :mel_len]` as `<item_id>.npy` (the files a
vocoder is fine-tuned on).  The batches come through the async device prefetcher; the device->host copy of step i
overlaps the forward of step i+1 (pinned staging buffer + a copy stream)."""
import itertools
from pathlib import Path
from typing import Any, Dict, Iterable, Union

import numpy as np
import torch

from .datapath import DevicePrefetcher


def create_gta_features(model: torch.nn.Module, train_set: Iterable[Dict[str, Any]],
                        val_set: Iterable[Dict[str, Any]], save_path: Union[Path, str]) -> int:
    """Returns the number of files written."""
    save_path = Path(save_path)
    save_path.mkdir(parents=True, exist_ok=True)
    model.eval()
    device = next(model.parameters()).device
    copy_stream = torch.cuda.Stream(device=device) if device.type == 'cuda' else None
    pending = None            # (host tensor, event, item ids, lens) of the previous batch
    written = 0

    def flush(p) -> int:
        host, ev, ids, lens = p
        if ev is not None:
            ev.synchronize()
        gta = host.numpy()
        for j, item_id in enumerate(ids):
            np.save(str(save_path / f'{item_id}.npy'), gta[j][:, :int(lens[j])], allow_pickle=False)
        return len(ids)

    for batch in DevicePrefetcher(itertools.chain(train_set, val_set), device):
        lens = [int(v) for v in batch['mel_len'].tolist()]       # host copy of the lengths (tiny)
        with torch.no_grad():
            pred = model(batch)
        post = pred['mel_post']
        if copy_stream is not None:
            host = torch.empty(post.shape, dtype=post.dtype).pin_memory()
            copy_stream.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(copy_stream):
                host.copy_(post, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            post.record_stream(copy_stream)
        else:
            host, ev = post.cpu(), None
        if pending is not None:
            written += flush(pending)
        pending = (host, ev, list(batch['item_id']), lens)
    if pending is not None:
        written += flush(pending)
    return written
