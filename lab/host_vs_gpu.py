"""Where is the HOST while the GPU runs a steady-state train step?  Every mark is recorded twice: on the stream the caller
is on (completes when the GPU gets there) and on an otherwise idle probe stream (completes at once: the host's position
in the GPU's clock).  A column pair that stays close = the GPU is waiting for the host there."""
import sys
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
probe = torch.cuda.Stream()
marks = []
on = [False]
def mark(name):
    if not on[0]:
        return
    g = torch.cuda.Event(enable_timing=True); h = torch.cuda.Event(enable_timing=True)
    g.record(); h.record(probe)
    marks.append((name, g, h))
orig_fwd = model.forward
def fwd(b):
    mark('FORWARD START'); r = orig_fwd(b); mark('forward issued'); return r
model.forward = fwd
orig_bw = torch.Tensor.backward
n = [0]
def bw(self, *a, **k):
    n[0] += 1; k_ = n[0]
    mark(f'backward call {k_}: before'); r = orig_bw(self, *a, **k); mark(f'backward call {k_}: returned'); return r
torch.Tensor.backward = bw
orig_opt = ts.optimizer_step
def opt():
    mark('before optimizer'); orig_opt(); mark('optimizer issued'); n[0] = 0
ts.optimizer_step = opt
for name in ('prenet', 'lstm', 'postnet'):
    mod = getattr(model, name)
    of = mod.forward
    def wrap(of=of, name=name):
        def f(*a, **k):
            mark('  before ' + name); r = of(*a, **k); mark('  ' + name + ' issued'); return r
        return f
    mod.forward = wrap()
NS = 12
for i in range(NS):
    batch['dur'].copy_(dur0)
    if i == NS - 3:
        on[0] = True
    ts.step(batch)
torch.cuda.synchronize()
starts = [i for i, m in enumerate(marks) if m[0] == 'FORWARD START']
for si, s in enumerate(starts):
    e = starts[si + 1] if si + 1 < len(starts) else len(marks)
    t0 = marks[s][2]
    print(f'--- step {si}:   host position | GPU (caller stream) reaches the mark   [ms after the host began the step]')
    for nm, g, h in marks[s:e]:
        print(f'{t0.elapsed_time(h):9.3f} | {t0.elapsed_time(g):9.3f}   {nm}')
