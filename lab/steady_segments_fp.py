"""Unprofiled STEADY-STATE seams of a train step (no sync before the measured steps: the host runs ahead as in the
bench): HIP events on the main stream at forward sub-seams and backward stages, plus the completion times of the two
side streams (weight gradients, predictors) right before the final joins."""
import sys, os
sys.path.insert(0, '.')
import torch
from forwardtacotron_amd import data, trainer as T, ops
from forwardtacotron_amd.fastpitch import FastPitch
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = FastPitch(**data.FASTPITCH_MODEL).to(dev); model.matmul_dtype = "bf16"
ts = T.TrainStep(model, lr=5e-5, train_cfg=dict(data.SINGLESPEAKER_TRAIN), gc_freeze=True)
batch = data.to_device(data.synthetic_batch(B=32, Tmax=128, n_mels=80, seed=0), dev)
dur0 = batch['dur'].clone()
marks = []
on = [False]
def mark(name, stream=None):
    if not on[0]:
        return
    e = torch.cuda.Event(enable_timing=True)
    e.record(stream) if stream is not None else e.record()
    marks.append((name, e))
orig_fwd = model.forward
def fwd(b):
    mark('FORWARD START'); r = orig_fwd(b); mark('forward end'); return r
model.forward = fwd
orig_bw = torch.Tensor.backward
n = [0]
def bw(self, *a, **k):
    r = orig_bw(self, *a, **k); n[0] += 1; mark(f'backward call {n[0]} issued (stream of the caller)'); return r
torch.Tensor.backward = bw
orig_flush = ops.flush_deferred
def flush():
    r = orig_flush(); mark('main: own work done (before joins)'); return r
ops.flush_deferred = flush
T.ops.flush_deferred = flush
orig_opt = ts.optimizer_step
def opt():
    mark('side: weight-gradient stream done', ts.wgrad_stream)
    try:
        mark('side: predictor stream done', model._side_stream(dev))
    except Exception as e:
        print('no predictor stream', e)
    mark('main: joined')
    orig_opt(); mark('clip + adam'); n[0] = 0
ts.optimizer_step = opt
for name in ('prenet', 'postnet', 'lr', 'dur_pred', 'pitch_pred', 'energy_pred'):
    mod = getattr(model, name)
    of = mod.forward
    def wrap(of=of, name=name):
        def f(*a, **k):
            mark('  before ' + name); r = of(*a, **k); mark('  ' + name); return r
        return f
    mod.forward = wrap()
NS = 12
for i in range(NS):
    batch['dur'].copy_(dur0)
    if i == NS - 3:
        on[0] = True
    ts.step(batch)
torch.cuda.synchronize()
starts = [i for i, (nm, _) in enumerate(marks) if nm == 'FORWARD START']
for si, s in enumerate(starts):
    e = starts[si + 1] if si + 1 < len(starts) else len(marks)
    t0 = marks[s][1]
    print(f'--- step {si} (next step starts at +{t0.elapsed_time(marks[e][1]) if e < len(marks) else float("nan"):.3f} ms)')
    rows = sorted(((t0.elapsed_time(ev), nm) for nm, ev in marks[s:e]))
    prev = 0.0
    for t, nm in rows:
        print(f'{t:8.3f} ms  (+{t - prev:6.3f})  {nm}')
        prev = t
