"""GPU time of each of 40 consecutive train steps (event pairs), to see whether step time drifts or has periodic outliers."""
import sys, time
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
import gc, os
N = int(os.environ.get("N", "120"))
if os.environ.get("NOGC") == "1":
    gc.disable()
if os.environ.get("FREEZE") == "1":
    gc.collect(); gc.freeze()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
host = []
torch.cuda.synchronize()
ev[0].record()
t0 = time.perf_counter()
for i in range(N):
    batch['dur'].copy_(dur0)
    ts.step(batch)
    ev[i + 1].record()
    host.append(time.perf_counter())
torch.cuda.synchronize()
t1 = time.perf_counter()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
print('outliers (>30 ms):', [(i, round(m, 1)) for i, m in enumerate(ms) if m > 30], 'gc counts', gc.get_count(), 'gc stats', [g['collections'] for g in gc.get_stats()])
print('wall per step over all', (t1 - t0) / N * 1e3, 'median', sorted(ms)[len(ms)//2])
