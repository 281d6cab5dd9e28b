"""Unprofiled main-stream timeline of a train step: HIP events at the seams (forward end, each backward stage, join, Adam)."""
import sys, os
sys.path.insert(0, '.')
import torch
from forwardtacotron_amd import data, trainer as T, ops
from forwardtacotron_amd.model import ForwardTacotron
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = ForwardTacotron(**data.SINGLESPEAKER_MODEL).to(dev)
ts = T.TrainStep(model, lr=5e-5, train_cfg=dict(data.SINGLESPEAKER_TRAIN))
batch = data.to_device(data.synthetic_batch(B=32, Tmax=128, n_mels=80, seed=0), dev)
dur0 = batch['dur'].clone()
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
# seams: forward (model.__call__), the three backward roots, optimizer_step
orig_fwd = model.forward
def fwd(b):
    mark('step start -> forward start'); r = orig_fwd(b); mark('forward'); return r
model.forward = fwd
orig_bw = torch.Tensor.backward
n = [0]
def bw(self, *a, **k):
    r = orig_bw(self, *a, **k); n[0] += 1; mark(f'backward stage {n[0]} (issued)'); return r
orig_opt = ts.optimizer_step
def opt():
    mark('joins'); orig_opt(); mark('clip + adam')
ts.optimizer_step = opt
# sub-seams inside the trunk
for name in ('prenet', 'lstm', 'postnet', 'lr'):
    mod = getattr(model, name)
    of = mod.forward
    def wrap(of=of, name=name):
        def f(*a, **k):
            mark('before ' + name); r = of(*a, **k); mark(name); return r
        return f
    mod.forward = wrap()
for i in range(8):
    batch['dur'].copy_(dur0)
    if i == 7:
        torch.cuda.synchronize(); marks.clear(); n[0] = 0
        torch.Tensor.backward = bw
        mark('t0')
    ts.step(batch)
mark('step end')
torch.cuda.synchronize()
t0 = marks[0][1]
prev = 0.0
for name, e in marks[1:]:
    t = t0.elapsed_time(e)
    print(f'{t:8.3f} ms  (+{t - prev:6.3f})  {name}')
    prev = t
