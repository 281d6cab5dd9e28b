"""Sanity run, not a benchmark: N train steps of the full-size ForwardTacotron on ONE synthetic batch (the bench's bs=32
batch) with the config's dropout -- the loss must fall, no recurrence may fault, the step time must stay flat.
    python tools/train_smoke.py [--steps 300] [--lr 1e-4]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--lr', type=float, default=1e-4)
    args = ap.parse_args()
    from forwardtacotron_amd import data, hip
    from forwardtacotron_amd.model import ForwardTacotron
    from forwardtacotron_amd.trainer import TrainStep
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = ForwardTacotron(**data.SINGLESPEAKER_MODEL).to(dev)
    ts = TrainStep(model, lr=args.lr, train_cfg=dict(data.SINGLESPEAKER_TRAIN), gc_freeze=True)
    batch = data.to_device(data.synthetic_batch(B=32, Tmax=128, n_mels=80, seed=0), dev)
    dur0 = batch['dur'].clone()
    hist = []
    t0 = time.time()
    for i in range(args.steps):
        batch['dur'].copy_(dur0)
        out = ts.step(batch)
        if i % 25 == 0 or i == args.steps - 1:
            torch.cuda.synchronize()
            rec = {k: float(out[k]) for k in ('loss', 'mel', 'mel_post', 'dur', 'pitch', 'energy', 'grad_norm')}
            rec['step'] = i
            rec['wall_s'] = round(time.time() - t0, 2)
            hist.append(rec)
            print(rec, flush=True)
    torch.cuda.synchronize()
    hip.check_rnn_status()
    ts.check()
    assert all(torch.isfinite(p).all() for p in model.parameters())
    assert hist[-1]['loss'] < 0.6 * hist[0]['loss'], (hist[0]['loss'], hist[-1]['loss'])
    print(f"ok: loss {hist[0]['loss']:.3f} -> {hist[-1]['loss']:.3f} in {args.steps} steps, "
          f"{(time.time() - t0) / args.steps * 1e3:.1f} ms/step wall incl. the periodic syncs; rnn launches {hip.rnn_counters()}")


if __name__ == '__main__':
    main()
