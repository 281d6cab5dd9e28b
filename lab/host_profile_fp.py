import cProfile, pstats, sys, io, torch
sys.path.insert(0, '.')
from forwardtacotron_amd import data
from forwardtacotron_amd.fastpitch import FastPitch
from forwardtacotron_amd.trainer import TrainStep
torch.manual_seed(0)
model = FastPitch(**data.FASTPITCH_MODEL).cuda(); model.matmul_dtype = 'bf16'
ts = TrainStep(model, lr=5e-5, train_cfg=dict(data.SINGLESPEAKER_TRAIN), gc_freeze=True)
batch = data.to_device(data.synthetic_batch(B=32, Tmax=128, n_mels=80, seed=0), 'cuda'); dur0 = batch['dur'].clone()
def step():
    batch['dur'].copy_(dur0); return ts.step(batch)
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])
