"""The N>1 path of the REAL train step on the GPU: two ranks (both on cuda:0 -- the box has one card -- talking over
gloo, which moves CUDA tensors through the host) run TrainStep with the gradient sink, the weight-gradient side stream
and the bucketed all-reduce on its communication stream.  Checked against a single-process computation of the same
thing: Adam on the clipped MEAN of the two ranks' gradients."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from helpers import TINY, TRAIN_CFG, load_npz, sub

pytestmark = pytest.mark.gpu
LR = 1e-3


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(rank):
    from oracle.ft_oracle import synthetic_batch
    return synthetic_batch(B=3, Tmax=9, n_mels=TINY['n_mels'], max_dur=5, seed=40 + rank)


def _model():
    from forwardtacotron_amd.model import ForwardTacotron
    M = load_npz('tiny_model.npz')
    m = ForwardTacotron(**TINY)
    m.load_state_dict({k: torch.as_tensor(v) for k, v in sub(M, 'sd/').items()})
    return m.cuda()


def _worker(rank, world, port, bucket_bytes, q):
    import torch.distributed as dist
    from forwardtacotron_amd.trainer import TrainStep
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    m = _model()
    ts = TrainStep(m, lr=LR, train_cfg=TRAIN_CFG, bucket_bytes=bucket_bytes)
    out = ts.step({k: v.cuda() for k, v in _batch(rank).items()})
    torch.cuda.synchronize()
    # numpy: pickled by value (torch tensors would travel as shared-memory handles that die with this process)
    q.put((rank, {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}, float(out['grad_norm']),
           len(ts.reducer.buckets)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('bucket_bytes', [4096, 24 << 20])
def test_train_step_world2_matches_mean_gradient(bucket_bytes):
    from forwardtacotron_amd.trainer import TrainStep
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bucket_bytes, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    sd0 = {k: torch.from_numpy(v) for k, v in res[0][1].items()}
    sd1 = {k: torch.from_numpy(v) for k, v in res[1][1].items()}
    differ = [k for k in sd0 if sd0[k].dtype.is_floating_point and 'running_' not in k and not torch.equal(sd0[k], sd1[k])]
    assert not differ, differ                                       # every rank applied the identical update
    if bucket_bytes == 4096:
        assert res[0][3] > 1                                        # really bucketed

    # single process: raw gradients of each rank's batch (lr = 0 leaves the weights alone), then Adam on the mean
    grads = []
    for r in range(2):
        m = _model()
        ts = TrainStep(m, lr=0.0, train_cfg=TRAIN_CFG)
        ts.step({k: v.cuda() for k, v in _batch(r).items()})
        grads.append({n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()})
    m = _model()
    p0 = {n: p.detach().cpu().clone() for n, p in m.named_parameters()}
    g = {n: 0.5 * (grads[0][n] + grads[1][n]) for n in p0}
    norm = torch.sqrt(sum((v.double() ** 2).sum() for v in g.values())).float()
    assert abs(float(norm) - res[0][2]) < 1e-4 * max(1.0, float(norm))
    coef = min(1.0, TRAIN_CFG['clip_grad_norm'] / (float(norm) + 1e-6))
    for n in p0:
        gc = g[n] * coef
        mhat = (0.1 * gc) / (1 - 0.9)
        vhat = (0.001 * gc * gc) / (1 - 0.999)
        want = p0[n] - LR * mhat / (vhat.sqrt() + 1e-8)
        live = g[n].abs() > 1e-6
        assert float((sd0[n][live] - want[live]).abs().max() if live.any() else 0.0) < 2e-5, n


def _rccl_worker(port, bucket_bytes, q):
    import torch.distributed as dist
    from forwardtacotron_amd.trainer import TrainStep
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['FT_DP_FORCE_COLLECTIVE'] = '1'
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    m = _model()
    ts = TrainStep(m, lr=LR, train_cfg=TRAIN_CFG, bucket_bytes=bucket_bytes)
    assert ts.reducer.active and ts.reducer.world == 1
    norms = []
    for i in range(2):
        out = ts.step({k: v.cuda() for k, v in _batch(i).items()})
        norms.append(float(out['grad_norm']))
    torch.cuda.synchronize()
    launched = sum(ts.reducer.launched)
    q.put(({k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}, norms, len(ts.reducer.buckets), launched,
           ts.reducer.comm_stream is not None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('bucket_bytes', [4096])
def test_train_step_over_rccl_one_rank_group(bucket_bytes):
    """The box has one card, so RCCL cannot be given two ranks; what CAN run for real is the whole N>1 code path over
    the 'nccl' backend (= RCCL) in a one-rank group with FT_DP_FORCE_COLLECTIVE=1: init_process_group(device_id),
    per-bucket ncclAllReduce launches on the communication stream chained behind the producer streams, the waits before
    clip + Adam.  A sum over one rank is the identity: two steps must equal the undistributed TrainStep bit for bit."""
    from forwardtacotron_amd.trainer import TrainStep
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), bucket_bytes, q))
    p.start()
    sd, norms, n_buckets, launched, had_comm_stream = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert n_buckets > 1 and launched == n_buckets and had_comm_stream
    m = _model()
    ts = TrainStep(m, lr=LR, train_cfg=TRAIN_CFG)
    want_norms = [float(ts.step({k: v.cuda() for k, v in _batch(i).items()})['grad_norm']) for i in range(2)]
    torch.cuda.synchronize()
    assert norms == want_norms
    for k, v in m.state_dict().items():
        assert torch.equal(v.cpu(), torch.from_numpy(sd[k])), k


# ---------------------------------------------------------------------------------------------------
# a recurrence fault on ONE rank must be acted on by EVERY rank
# ---------------------------------------------------------------------------------------------------
def _fault_worker(rank, world, port, q):
    import torch.distributed as dist
    from forwardtacotron_amd import _lib, hip as H
    from forwardtacotron_amd.model import ForwardTacotron
    from forwardtacotron_amd.trainer import TrainStep
    from oracle.ft_oracle import synthetic_batch
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    cfg = dict(TINY, rnn_dims=32, postnet_dims=32, durpred_rnn_dims=16)    # persistent-eligible recurrences
    torch.manual_seed(3)
    m = ForwardTacotron(**cfg).cuda()
    ts = TrainStep(m, lr=1e-3, train_cfg=TRAIN_CFG, bucket_bytes=4096)
    batch = synthetic_batch(B=4, Tmax=12, n_mels=10, max_dur=5, seed=50 + rank)
    dev = lambda: {k: v.clone().cuda() for k, v in batch.items()}
    res = {'rank': rank}
    ts.step(dev())
    torch.cuda.synchronize()
    before = (ts.flat.flat.clone(), ts.exp_avg.clone(), ts.exp_avg_sq.clone(), ts.bufs.stats.clone(), ts.bufs.counts.clone())
    n0 = H.rnn_counters()[0]
    if rank == 1:
        _lib.lib().ft_rnn_set_max_spins(-1)                  # every poll of THIS rank's persistent recurrences fails
    try:
        out = ts.step(dev())
        torch.cuda.synchronize()
    finally:
        _lib.lib().ft_rnn_set_max_spins(0)
    res['persistent_ran'] = H.rnn_counters()[0] > n0
    res['flag'] = float(out['rnn_fault'])
    res['norm_nan'] = bool(torch.isnan(out['grad_norm']))
    after = (ts.flat.flat, ts.exp_avg, ts.exp_avg_sq, ts.bufs.stats, ts.bufs.counts)
    res['untouched'] = all(torch.equal(a, b) for a, b in zip(before, after))
    out = ts.step(dev())                                      # the word is sticky on rank 1: skipped again, everywhere
    torch.cuda.synchronize()
    res['flag2'] = float(out['rnn_fault'])
    res['untouched2'] = all(torch.equal(a, b) for a, b in zip(before, after))
    try:
        ts.step(dev())                                        # the faulted step's flag is two steps old: every rank raises
        res['raised'] = None
    except _lib.FtError as e:
        res['raised'] = str(e)
    res['opt_step'], res['skipped'] = ts.opt_step, ts.skipped_steps
    norms = []
    for _ in range(2):                                        # ... and both carry on, identically
        out = ts.step(dev())
        norms.append(float(out['grad_norm']))
    ts.check()
    torch.cuda.synchronize()
    res['norms'] = norms
    res['flag3'] = float(out['rnn_fault'])
    res['sd'] = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    res['moments'] = (ts.exp_avg.cpu().numpy(), ts.exp_avg_sq.cpu().numpy())
    q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_recurrence_fault_on_one_rank_is_global():
    """Two ranks (one GPU, gloo), rank 1's persistent recurrences forced to time out in the second step.  Rank 1's garbage
    gradient is summed into both ranks' buckets; the fault lane (all-reduced behind the buckets) makes BOTH ranks skip the
    update on the device: parameters, Adam moments, BatchNorm statistics and counters bit-unchanged on both, rnn_fault = 1
    on the faulting rank and 2 on the other; both raise at the start of the same later step, both rewind Adam's step
    count alike, and the steps after that leave both ranks with bit-identical parameters and moments."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fault_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t['rank'])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    r0, r1 = res
    assert r1['persistent_ran'], 'the persistent form must have run on the faulting rank (else this tests nothing)'
    assert r1['flag'] == 1.0 and r0['flag'] == 2.0
    for r in res:
        assert r['norm_nan'] and r['untouched'] and r['untouched2'], r['rank']
        assert r['flag2'] != 0.0
        assert r['raised'] is not None and 'timed out' in r['raised'], r['rank']
        assert r['opt_step'] == 1 and r['skipped'] == 2, (r['opt_step'], r['skipped'])   # rewound to the last real update
        assert r['flag3'] == 0.0
    assert 'ANOTHER rank' in r0['raised'] and 'ANOTHER rank' not in r1['raised']
    assert r0['norms'] == r1['norms']
    for k in r0['sd']:
        if r0['sd'][k].dtype.kind == 'f' and 'running_' in k:
            continue                                            # BatchNorm statistics are per rank (no SyncBN)
        assert (r0['sd'][k] == r1['sd'][k]).all(), k
    assert (r0['moments'][0] == r1['moments'][0]).all() and (r0['moments'][1] == r1['moments'][1]).all()
