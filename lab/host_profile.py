"""Host-side cost of one train step: enqueue time with the GPU drained before (so nothing blocks but the
LengthRegulator's size read), then a cProfile of 5 steps."""
import cProfile, pstats, sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from forwardtacotron_amd import data
from forwardtacotron_amd.model import ForwardTacotron
from forwardtacotron_amd.trainer import TrainStep
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = ForwardTacotron(**data.SINGLESPEAKER_MODEL).to(dev)
ts = TrainStep(model, lr=5e-5, train_cfg=dict(data.SINGLESPEAKER_TRAIN))
batch = data.to_device(data.synthetic_batch(B=32, Tmax=128, n_mels=80, seed=0), dev)
dur0 = batch['dur'].clone()
def one():
    batch['dur'].copy_(dur0)
    return ts.step(batch)
for _ in range(5): one()
torch.cuda.synchronize()
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); one(); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'host enqueue {1e3*(t1-t0):.2f} ms, until drained {1e3*(t2-t0):.2f} ms')
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for _ in range(5): one()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(35)
st.sort_stats('cumulative').print_stats(45)
