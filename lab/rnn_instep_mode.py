"""Do the recurrences of a real train step run the XCD-local hand-off?  (mode counts per step) -- and does a GRU-128
forward launched right behind a large GEMM (as in the step) take longer than on an idle chip?"""
import sys, torch
sys.path.insert(0, '.')
from forwardtacotron_amd import data, hip as H
from forwardtacotron_amd.model import ForwardTacotron
from forwardtacotron_amd.trainer import TrainStep
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = ForwardTacotron(**data.SINGLESPEAKER_MODEL).to(dev)
ts = TrainStep(model, lr=5e-5, train_cfg=dict(data.SINGLESPEAKER_TRAIN))
batch = data.to_device(data.synthetic_batch(B=32, Tmax=128, n_mels=80, seed=0), dev)
dur0 = batch['dur'].clone()
for i in range(6):
    batch['dur'].copy_(dur0)
    m0 = H.rnn_mode_counts()
    ts.step(batch)
    torch.cuda.synchronize()
    m1 = H.rnn_mode_counts()
    print(f'step {i}: groups XCD-local {m1[0] - m0[0]}, agent-scope {m1[1] - m0[1]}', flush=True)
T, B, Hh = 841, 32, 128
xp = torch.randn(T, B, 6 * Hh, device=dev) * 0.1
whh = [torch.randn(3 * Hh, Hh, device=dev) * 0.03 for _ in range(2)]
bhh = [torch.zeros(3 * Hh, device=dev) for _ in range(2)]
a = torch.randn(26912, 256, device=dev); w = torch.randn(2048, 256, device=dev) * 0.05
def gru():
    return H.gru_fwd(xp, whh[0], whh[1], bhh[0], bhh[1], Hh, True)
def timed(pre):
    torch.cuda.synchronize()
    for _ in range(pre):
        H.linear_fwd(a, w)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); gru(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / T
gru(); torch.cuda.synchronize()
for pre in (0, 0, 4, 4, 16, 16, 64, 64, 0, 0):
    print(f'GRU-128 forward behind {pre:2d} GEMMs (26912x256x2048): {timed(pre):.2f} us/step', flush=True)
# hypothesis: in the step the GRU's per-step operands (xp forward; dout / gates / out backward) come from HBM, in the
# timing loops above they sit in the 256 MB Infinity Cache from the previous call.  Evict with a 2 GB copy in front.
big_a = torch.empty(512 << 20, device=dev, dtype=torch.float32)        # 2 GB
big_b = torch.empty_like(big_a)
def timed2(evict, fn):
    torch.cuda.synchronize()
    if evict:
        big_b.copy_(big_a)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); r = fn(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / T, r
out, gates = gru()
wt = [H.transpose2d(w_) for w_ in whh]
dout = torch.randn(T, B, 2 * Hh, device=dev) * 0.1
def gru_b():
    return H.gru_bwd(dout, out, gates, wt[0], wt[1], Hh)
gru_b(); torch.cuda.synchronize()
for evict in (0, 1, 0, 1, 1):
    print(f'GRU-128 evict={evict}: forward {timed2(evict, gru)[0]:.2f} us/step, backward {timed2(evict, gru_b)[0]:.2f} us/step', flush=True)
# is it the stream?  TrainStep runs the step on a HIGH-priority stream
hp = torch.cuda.Stream(priority=-1)
lp = torch.cuda.Stream()
for name, st in (('high-priority stream', hp), ('plain side stream', lp), ('high-priority stream', hp)):
    with torch.cuda.stream(st):
        gru(); gru_b(); torch.cuda.synchronize()
        print(f'GRU-128 on a {name}: forward {timed2(0, gru)[0]:.2f} us/step, backward {timed2(0, gru_b)[0]:.2f} us/step', flush=True)
# the real thing: the postnet GRU call of a train step, timed in the step and replayed alone with the same tensors
calls = []
orig = H.gru_fwd
def spy(xp_, *a_, **k_):
    if xp_.shape[0] > 200:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); r = orig(xp_, *a_, **k_); e.record()
        calls.append((s, e, xp_, a_, k_))
        return r
    return orig(xp_, *a_, **k_)
H.gru_fwd = spy
import forwardtacotron_amd.ops as O
for i in range(3):
    batch['dur'].copy_(dur0)
    ts.step(batch)
torch.cuda.synchronize()
H.gru_fwd = orig
for s, e, xp_, a_, k_ in calls:
    print(f'postnet GRU in the step: T={xp_.shape[0]} {s.elapsed_time(e) * 1e3 / xp_.shape[0]:.2f} us/step', flush=True)
s, e, xp_, a_, k_ = calls[-1]
for st in (hp, None):
    ctx = torch.cuda.stream(st) if st is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        orig(xp_, *a_, **k_); torch.cuda.synchronize()
        s2, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s2.record(); orig(xp_, *a_, **k_); e2.record(); torch.cuda.synchronize()
        print(f'  the same call replayed alone ({"high-priority" if st is not None else "default"} stream): {s2.elapsed_time(e2) * 1e3 / xp_.shape[0]:.2f} us/step', flush=True)
# which argument makes the real call slower than the synthetic one?
def t_call(xp__, a__):
    orig(xp__, *a__, **k_); torch.cuda.synchronize()
    s3, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s3.record(); orig(xp__, *a__, **k_); e3.record(); torch.cuda.synchronize()
    return s3.elapsed_time(e3) * 1e3 / xp__.shape[0]
print('arg kinds:', [type(v).__name__ + (str(tuple(v.shape)) if torch.is_tensor(v) else repr(v)) for v in a_], k_)
print(f'real xp stats: mean {xp_.mean().item():.3f} std {xp_.std().item():.3f} absmax {xp_.abs().max().item():.3f}; '
      f'whh std {a_[0].std().item():.4f}; bhh absmax {a_[2].abs().max().item():.4f}', flush=True)
rx = torch.randn_like(xp_) * 0.1
print(f'real everything            : {t_call(xp_, a_):.2f}')
print(f'real xp cloned             : {t_call(xp_.clone(), a_):.2f}')
print(f'random small xp, real w    : {t_call(rx, a_):.2f}')
print(f'random xp with real std    : {t_call(torch.randn_like(xp_) * xp_.std() + xp_.mean(), a_):.2f}')
a_rw = (torch.randn_like(a_[0]) * 0.03, torch.randn_like(a_[1]) * 0.03, torch.zeros_like(a_[2]), torch.zeros_like(a_[3])) + tuple(a_[4:])
print(f'real xp, random w zero bias: {t_call(xp_, a_rw):.2f}')
print(f'random xp, random w        : {t_call(rx, a_rw):.2f}')
a_cl = tuple(v.clone() if torch.is_tensor(v) else v for v in a_)
print(f'real xp, real w cloned     : {t_call(xp_, a_cl):.2f}', flush=True)
