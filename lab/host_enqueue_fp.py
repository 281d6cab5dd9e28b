"""Host enqueue time of a FastPitch train step (GPU drained in front) against the time until the GPU has drained."""
import sys, time, torch
sys.path.insert(0, '.')
from forwardtacotron_amd import data
from forwardtacotron_amd.fastpitch import FastPitch
from forwardtacotron_amd.trainer import TrainStep
dev = torch.device('cuda', 0)
for dtype in ('fp32', 'bf16'):
    torch.manual_seed(0)
    model = FastPitch(**data.FASTPITCH_MODEL).to(dev)
    model.matmul_dtype = dtype
    ts = TrainStep(model, lr=5e-5, train_cfg=dict(data.SINGLESPEAKER_TRAIN))
    batch = data.to_device(data.synthetic_batch(B=32, Tmax=128, n_mels=80, seed=0), dev)
    dur0 = batch['dur'].clone()
    def one():
        batch['dur'].copy_(dur0)
        return ts.step(batch)
    for _ in range(5): one()
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); one(); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f'{dtype}: host enqueue {1e3*(t1-t0):.2f} ms, until drained {1e3*(t2-t0):.2f} ms', flush=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): one()
    torch.cuda.synchronize(); print(f'{dtype}: steady {1e2*(time.perf_counter()-t0):.2f} ms/step', flush=True)
