"""Single process, pretend world=2: which parameters notify the reducer how many times, and when is a bucket launched
relative to its members' notifications?"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import collections
import torch
import torch.distributed as dist

calls = []


class W:
    def wait(self):
        pass


dist.is_initialized = lambda: True
dist.get_world_size = lambda group=None: 2
dist.all_reduce = lambda t, op=None, group=None, async_op=False: (calls.append(('AR', t.data_ptr(), t.numel())), W())[1]

from test_gpu_parallel import _model, _batch
from helpers import TRAIN_CFG
from forwardtacotron_amd.trainer import TrainStep
from forwardtacotron_amd import parallel

m = _model()
ts = TrainStep(m, lr=1e-3, train_cfg=TRAIN_CFG, bucket_bytes=4096)
red = ts.reducer
cnt = collections.Counter()
orig = red.notify


def notify(j):
    cnt[j] += 1
    calls.append(('N', j))
    orig(j)


red.notify = notify
ts.sink.on_write = notify
ts.step({k: v.cuda() for k, v in _batch(0).items()})
torch.cuda.synchronize()
names = ts.flat.names
print('params', len(names), 'buckets', len(red.buckets))
print('notified != 1:', [(names[j], c) for j, c in cnt.items() if c != 1][:20])
print('never notified:', [names[j] for j in range(len(names)) if cnt[j] == 0][:40])
# bucket launch position vs last notify of its members
seen = set()
base = ts.flat.grad.data_ptr()
for c in calls:
    if c[0] == 'N':
        seen.add(c[1])
    else:
        lo = (c[1] - base) // 4
        b = [i for i, (l, h) in enumerate(red.buckets) if l == lo][0]
        members = [j for j in range(len(names)) if red.param_bucket[j] == b]
        missing = [names[j] for j in members if j not in seen]
        if missing:
            print('bucket', b, 'launched before', missing[:6])
