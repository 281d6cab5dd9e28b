"""GPU time of a FastPitch step with the host far ahead: a long sleep kernel in front, events around the step"""
import sys, torch
sys.path.insert(0, '.')
from forwardtacotron_amd import data
from forwardtacotron_amd.fastpitch import FastPitch
from forwardtacotron_amd.trainer import TrainStep
torch.manual_seed(0)
model = FastPitch(**data.FASTPITCH_MODEL).cuda(); model.matmul_dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
ts = TrainStep(model, lr=5e-5, train_cfg=dict(data.SINGLESPEAKER_TRAIN), gc_freeze=True)
batch = data.to_device(data.synthetic_batch(B=32, Tmax=128, n_mels=80, seed=0), 'cuda'); dur0 = batch['dur'].clone()
def step():
    batch['dur'].copy_(dur0); return ts.step(batch)
for _ in range(5): step()
torch.cuda.synchronize()
for _ in range(3):
    torch.cuda._sleep(int(60e6 * 2.1))          # ~60 ms: the host enqueues the whole step meanwhile (LR's .item() aside)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); step(); e.record(); torch.cuda.synchronize()
    print(f'{model.matmul_dtype}: GPU time of one step with the host ahead: {s.elapsed_time(e):.2f} ms (includes the wait for the LengthRegulator size read-back)')
