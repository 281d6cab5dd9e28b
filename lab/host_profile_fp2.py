"""where the host time of a FastPitch bf16 step goes: wall time inside selected Python functions (any thread)"""
import sys, time, functools, collections, torch
sys.path.insert(0, '.')
from forwardtacotron_amd import data, hip as H, ops, _lib, fastpitch as FP
from forwardtacotron_amd.trainer import TrainStep
acc = collections.defaultdict(lambda: [0, 0.0])
def wrap(mod, name):
    f = getattr(mod, name)
    @functools.wraps(f)
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            e = acc[f'{mod.__name__.split(".")[-1]}.{name}']; e[0] += 1; e[1] += time.perf_counter() - t0
    setattr(mod, name, g)
for n in ('call', 'query'): wrap(_lib, n)
for n in ('linear_fwd', 'linear_bwd_data', 'linear_bwd_weight_raw', 'colsum_raw', 'colsum2_raw', 'workspace', 'attn_fwd', 'attn_bwd',
          'conv1d_bwd_data_raw', 'conv1d_bwd_weight_raw', 'conv_pack_weight', 'dropout', '_stream'): wrap(H, n)
for n in ('_emit', '_emit_multi', '_side_launch', '_sink_done'): wrap(ops, n)
for n in ('mha_fwd', 'mha_bwd', 'addln_fwd', 'addln_bwd', 'convbias_fwd', 'convbias_bwd'): wrap(FP, n)
FP._emit, FP._emit_multi = ops._emit, ops._emit_multi
orig_empty, orig_empty_like = torch.empty, torch.empty_like
def te(*a, **k):
    t0 = time.perf_counter(); r = orig_empty(*a, **k); e = acc['torch.empty']; e[0] += 1; e[1] += time.perf_counter() - t0; return r
def tel(*a, **k):
    t0 = time.perf_counter(); r = orig_empty_like(*a, **k); e = acc['torch.empty_like']; e[0] += 1; e[1] += time.perf_counter() - t0; return r
torch.empty, torch.empty_like = te, tel
torch.manual_seed(0)
model = FP.FastPitch(**data.FASTPITCH_MODEL).cuda(); model.matmul_dtype = 'bf16'
ts = TrainStep(model, lr=5e-5, train_cfg=dict(data.SINGLESPEAKER_TRAIN), gc_freeze=True)
batch = data.to_device(data.synthetic_batch(B=32, Tmax=128, n_mels=80, seed=0), 'cuda'); dur0 = batch['dur'].clone()
def step():
    batch['dur'].copy_(dur0); return ts.step(batch)
for _ in range(5): step()
torch.cuda.synchronize(); acc.clear()
N = 5
t0 = time.perf_counter()
for _ in range(N): step()
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f'host {1e3 * (t1 - t0) / N:.2f} ms/step')
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f'{k:28s} {n / N:7.1f} calls/step {1e3 * t / N:7.3f} ms/step  {1e6 * t / max(n, 1):6.1f} us/call')
