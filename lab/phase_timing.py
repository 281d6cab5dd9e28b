"""Unprofiled phase timing of the train step with HIP events on the main stream:
forward | backward (main stream) | wait for the weight-gradient / predictor streams | clip + Adam."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from forwardtacotron_amd import data, ops, hip as H
from forwardtacotron_amd.model import ForwardTacotron
from forwardtacotron_amd.trainer import TrainStep
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = ForwardTacotron(**data.SINGLESPEAKER_MODEL).to(dev)
ts = TrainStep(model, lr=5e-5, train_cfg=dict(data.SINGLESPEAKER_TRAIN))
batch = data.to_device(data.synthetic_batch(B=32, Tmax=128, n_mels=80, seed=0), dev)
dur0 = batch['dur'].clone()
ev = lambda: torch.cuda.Event(enable_timing=True)

def step(marks):
    batch['dur'].copy_(dur0)
    e = [ev() for _ in range(6)]
    e[0].record()
    pt, et = batch['pitch'].detach().clone(), batch['energy'].detach().clone()
    packs = ts._weight_packs(); packs.refresh(); H.pack_cache = packs
    pred = model(batch)
    L = ts.losses(pred, batch, pt, et)
    e[1].record()
    ts.flat.zero_grad(); ts.reducer.start(); ts.sink.begin_step(); ops.set_grad_sink(ts.sink)
    L['loss'].backward()
    ops.set_grad_sink(None); H.pack_cache = None
    e[2].record()
    torch.cuda.current_stream().wait_stream(ts.wgrad_stream)
    e[3].record()
    ts.reducer.finish(); ts.optimizer_step()
    e[4].record()
    marks.append(e)

model.train(); ts.flat.attach() if not ts.flat.attached() else None
for _ in range(5): step([])
marks = []
torch.cuda.synchronize()
for _ in range(10): step(marks)
torch.cuda.synchronize()
names = ['forward+loss', 'backward(main)', 'wait wgrad stream', 'clip+adam']
tot = 0
for i, n in enumerate(names):
    ms = sum(m[i].elapsed_time(m[i + 1]) for m in marks) / len(marks)
    tot += ms
    print(f'{n:20s} {ms:7.3f} ms')
print(f'{"sum":20s} {tot:7.3f} ms;  step-to-step {sum(marks[i][0].elapsed_time(marks[i+1][0]) for i in range(9))/9:.3f} ms')
