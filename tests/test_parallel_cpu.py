"""CPU, world_size 2, gloo: the flat-parameter layout and the bucketed, hook-driven gradient all-reduce
(forwardtacotron_amd/parallel.py) -- the N>1 path of the train step -- on a small stand-in module."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from forwardtacotron_amd.parallel import BucketedAllReduce, FlatParams


class Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1d_bank = torch.nn.ModuleList()
        for _ in range(3):
            blk = torch.nn.Module()
            blk.bnorm = torch.nn.BatchNorm1d(5)
            blk.lin = torch.nn.Linear(5, 5)
            self.conv1d_bank.append(blk)
        self.head = torch.nn.Linear(5, 3)
        self.unused = torch.nn.Linear(2, 2)

    def forward(self, x):
        for blk in self.conv1d_bank:
            x = blk.lin(x) * blk.bnorm.weight + blk.bnorm.bias
        return self.head(x)


def test_flat_params_layout_and_views():
    torch.manual_seed(0)
    m = Toy()
    ref = {n: p.detach().clone() for n, p in m.named_parameters()}
    f = FlatParams(m)
    for n, p in m.named_parameters():
        assert torch.equal(p.detach(), ref[n]), n              # values preserved
        assert p.grad is not None and p.grad.shape == p.shape
    # bank BatchNorm weights are contiguous, in member order
    w = [m.conv1d_bank[i].bnorm.weight for i in range(3)]
    assert w[1].data_ptr() == w[0].data_ptr() + 5 * 4 and w[2].data_ptr() == w[0].data_ptr() + 10 * 4
    assert f.attached()
    m(torch.randn(4, 5)).sum().backward()
    assert float(f.grad.abs().sum()) > 0
    f.zero_grad()
    assert float(f.grad.abs().sum()) == 0 and f.attached()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, bucket_bytes, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(0)
    m = Toy()
    f = FlatParams(m)
    red = BucketedAllReduce(f, bucket_bytes=bucket_bytes)
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(6, 5, generator=g)
    for it in range(2):                       # two backward passes: hooks re-arm correctly
        f.zero_grad()
        red.start()
        m(x).pow(2).sum().backward()
        red.finish()
    # reference: gradients of both ranks' losses, summed
    torch.manual_seed(0)
    m2 = Toy()
    tot = None
    for r in range(world):
        xr = torch.randn(6, 5, generator=torch.Generator().manual_seed(100 + r))
        for p in m2.parameters():
            p.grad = None
        m2(xr).pow(2).sum().backward()
        gr = {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in m2.named_parameters()}
        tot = gr if tot is None else {n: tot[n] + gr[n] for n in gr}
    worst = max(float((p.grad - tot[n]).abs().max()) for n, p in m.named_parameters())
    q.put((rank, worst, len(red.buckets)))
    dist.destroy_process_group()


@pytest.mark.parametrize('bucket_bytes', [64, 1 << 20])
def test_bucketed_allreduce_world2_gloo(bucket_bytes):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bucket_bytes, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, worst, nb in res:
        assert worst < 1e-5, (rank, worst)
    if bucket_bytes == 64:
        assert res[0][2] > 1
