"""Host time of the three backward stages of TrainStep (is something in the predictors' stage blocking the host?)."""
import sys, time, os
sys.path.insert(0, '.')
import torch
from forwardtacotron_amd import data, trainer as T
from forwardtacotron_amd.model import ForwardTacotron
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = ForwardTacotron(**data.SINGLESPEAKER_MODEL).to(dev)
ts = T.TrainStep(model, lr=5e-5, train_cfg=dict(data.SINGLESPEAKER_TRAIN))
batch = data.to_device(data.synthetic_batch(B=32, Tmax=128, n_mels=80, seed=0), dev)
dur0 = batch['dur'].clone()
orig = torch.Tensor.backward
marks = []
def timed(self, *a, **k):
    t0 = time.perf_counter()
    r = orig(self, *a, **k)
    marks.append(time.perf_counter() - t0)
    return r
for i in range(8):
    batch['dur'].copy_(dur0)
    if i == 5:
        torch.cuda.synchronize()
        torch.Tensor.backward = timed
    t0 = time.perf_counter()
    ts.step(batch)
    t1 = time.perf_counter()
    if i >= 5:
        print('step host ms', round((t1 - t0) * 1e3, 2), 'backward calls ms', [round(m * 1e3, 2) for m in marks]); marks.clear()
torch.cuda.synchronize()
