"""Data-parallel plumbing for the train step (new work: the reference is single-device, train_forward.py:70).

One process per GPU; gradients live in ONE flat fp32 buffer (parameters are views into a flat parameter
buffer, laid out so that the tensors a fused kernel wants adjacent ARE adjacent).  The flat gradient is
cut into buckets; a post-accumulate hook launches an asynchronous sum all-reduce (RCCL over xGMI for the
"nccl" backend, gloo on CPU in tests) as soon as every gradient of a bucket has been produced, so the
exchange overlaps the remaining backward (the 841-step recurrences dominate).
Nothing here is model specific, which is what lets the CPU/gloo tests exercise the N>1 path.
"""
import os
import re
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def _bank_group_key(name: str) -> Optional[Tuple[str, str, int]]:
    """conv1d_bank.{i}.bnorm.{weight,bias} members must be contiguous (one BatchNorm kernel per bank)."""
    m = re.match(r'^(.*conv1d_bank)\.(\d+)\.bnorm\.(weight|bias)$', name)
    return (m.group(1), m.group(3), int(m.group(2))) if m else None


class FlatParams:
    """Re-homes every parameter of `module` into one flat buffer (p.data becomes a view) and gives each
    parameter a .grad view into a flat gradient buffer.  Alignment: every tensor starts on a 16-byte
    boundary unless it belongs to a contiguity group (bank BatchNorm weights), which is packed densely."""

    def __init__(self, module: torch.nn.Module, group_key: Callable[[str], Optional[tuple]] = _bank_group_key):
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        groups: Dict[tuple, List[Tuple[int, str, torch.nn.Parameter]]] = {}
        rest = []
        for n, p in named:
            k = group_key(n)
            if k is None:
                rest.append((n, p))
            else:
                groups.setdefault(k[:2], []).append((k[2], n, p))
        order: List[Tuple[str, torch.nn.Parameter, bool]] = []      # (name, param, dense-follow)
        for key in sorted(groups):
            members = sorted(groups[key], key=lambda t: t[0])
            for j, (_, n, p) in enumerate(members):
                order.append((n, p, j > 0))
        for n, p in rest:
            order.append((n, p, False))
        self.names, self.params, self.offsets = [], [], []
        off = 0
        for n, p, dense in order:
            if not dense:
                off = (off + 3) // 4 * 4
            self.names.append(n)
            self.params.append(p)
            self.offsets.append(off)
            off += p.numel()
        self.total = (off + 3) // 4 * 4
        dev, dt = order[0][1].device, order[0][1].dtype
        self.flat = torch.zeros(self.total, device=dev, dtype=dt)
        self.grad = torch.zeros(self.total, device=dev, dtype=dt)
        for p, o in zip(self.params, self.offsets):
            self.flat[o:o + p.numel()].copy_(p.detach().reshape(-1))
        self.attach()

    def attach(self) -> None:
        """(Re)points p.data / p.grad at the flat buffers (call again if something replaced them)."""
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            p.data = self.flat[o:o + n].view(p.shape)
            p.grad = self.grad[o:o + n].view(p.shape)

    def attached(self) -> bool:
        p, o = self.params[-1], self.offsets[-1]
        return (p.data_ptr() == self.flat.data_ptr() + 4 * o and p.grad is not None
                and p.grad.data_ptr() == self.grad.data_ptr() + 4 * o)

    def zero_grad(self) -> None:
        self.grad.zero_()


class FlatBuffers:
    """Re-homes the buffers a TRAINING forward pass updates in place -- every BatchNorm's running_mean / running_var
    (fp32) and num_batches_tracked plus the model's `step` (int64; forward_tacotron.py:101,126-127) -- into one flat
    tensor per dtype, so that a step can snapshot them with two copies and a faulted step (ft_clip_grad_norm's record)
    can put them back on the device with two guarded launches: a recurrence fault then leaves NO trace in the model, not
    only none in the parameters.  Order: all running_means in module order, then all running_vars -- the members of a
    conv bank stay adjacent, which CBHG._bank_flat relies on -- then the counters in module order, `step` last (the
    layout model._bump_batchnorm_counters keeps itself).  load_state_dict copies in place and keeps the views;
    .to() / .cuda() break them: attach() again (TrainStep does)."""

    def __init__(self, module: torch.nn.Module):
        self.module = module
        bns = [m for m in module.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)
               and m.running_mean is not None]
        self.bns = bns
        dev = next(module.parameters()).device
        fl = [b.running_mean for b in bns] + [b.running_var for b in bns]
        self.f_sizes = [t.numel() for t in fl]
        self.stats = (torch.cat([t.detach().reshape(-1).float() for t in fl]) if fl
                      else torch.zeros(0)).to(dev).contiguous()
        cnt = [b.num_batches_tracked.detach().reshape(1) for b in bns]
        self.has_step = isinstance(getattr(module, 'step', None), torch.Tensor) and module.step.dtype == torch.int64
        if self.has_step:
            cnt.append(module.step.detach().reshape(1))
        self.counts = (torch.cat(cnt) if cnt else torch.zeros(0, dtype=torch.int64)).to(dev).contiguous()
        self.stats_snap = torch.empty_like(self.stats)
        self.counts_snap = torch.empty_like(self.counts)
        self.attach()

    def attach(self) -> None:
        n, off = len(self.bns), 0
        for j, name in enumerate(('running_mean', 'running_var')):
            for i, b in enumerate(self.bns):
                k = self.f_sizes[j * n + i]
                b._buffers[name] = self.stats[off:off + k].view(getattr(b, name).shape)
                off += k
        for i, b in enumerate(self.bns):
            b._buffers['num_batches_tracked'] = self.counts[i]
        if self.has_step:
            self.module._buffers['step'] = self.counts[n:n + 1]
        if hasattr(self.module, '_nbt_flat') and n:
            self.module._nbt_flat = self.counts[:n]

    def attached(self) -> bool:
        if self.has_step and self.module.step.data_ptr() != self.counts.data_ptr() + 8 * len(self.bns):
            return False
        return not self.bns or (self.bns[0].running_mean.data_ptr() == self.stats.data_ptr()
                                and self.bns[-1].num_batches_tracked.data_ptr()
                                == self.counts.data_ptr() + 8 * (len(self.bns) - 1))

    def snapshot(self) -> None:
        """values before this step's forward (two device copies on the current stream)"""
        if self.stats.numel():
            self.stats_snap.copy_(self.stats)
        if self.counts.numel():
            self.counts_snap.copy_(self.counts)


class BucketedAllReduce:
    """Sum all-reduce of a flat gradient buffer in buckets, overlapped with backward.

    Buckets are contiguous ranges of the flat buffer filled in the order gradients become ready
    (reverse registration order is a good prior: post_proj / postnet first).  `start()` arms the hooks'
    counters for one backward; `finish()` waits (stream-level) for all launched collectives and launches
    any bucket whose hooks never fired (unused parameters keep their zero gradient)."""

    def __init__(self, flat: FlatParams, process_group=None, bucket_bytes: int = 24 << 20, lane_fill=None):
        """lane_fill(lane): writes this rank's fault state (1.0 / 0.0) into lane[0] on the current stream.  With it, finish()
        sum-all-reduces a 4-float FAULT LANE behind the last bucket: a rank whose recurrence faulted has already summed
        its garbage gradient into every rank's buckets, so every rank must skip the update alike (the optimizer reads the
        lane, ft_clip_grad_norm).  One 16-byte collective per step, on the communication stream, after the buckets."""
        self.flat = flat
        self.pg = process_group
        self.lane_fill = lane_fill
        self.lane = None
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # FT_DP_FORCE_COLLECTIVE=1: a one-rank group still runs every bucket through the backend (a sum over one rank is
        # the identity) -- lets a one-GPU box execute the RCCL launch / communication-stream path for real
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get('FT_DP_FORCE_COLLECTIVE') == '1')
        # bucket boundaries over the flat layout, walking parameters from the END (first grads ready)
        self.buckets: List[Tuple[int, int]] = []
        self.param_bucket: List[int] = [0] * len(flat.params)
        hi = flat.total
        lo_idx = len(flat.params)
        cur_members = []
        limit = max(bucket_bytes // 4, 1)
        for i in range(len(flat.params) - 1, -1, -1):
            cur_members.append(i)
            start = flat.offsets[i]
            if hi - start >= limit or i == 0:
                if i == 0:
                    start = 0
                bi = len(self.buckets)
                self.buckets.append((start, hi))
                for j in cur_members:
                    self.param_bucket[j] = bi
                cur_members = []
                hi = start
        self.bucket_size = [0] * len(self.buckets)
        for j, b in enumerate(self.param_bucket):
            self.bucket_size[b] += 1
        self.pending = list(self.bucket_size)
        self.launched = [False] * len(self.buckets)
        self.works = []
        self.armed = False
        self.streams = set()
        self.seen = set()
        self.held = None                # set of parameter indices a gradient sink still owes (ops.GradSink.held)
        self.comm_stream = None
        self._hooks = []
        if not self.active:
            self.lane_fill = None           # (single rank, nothing forced through a backend: no lane)
        if self.active:
            for j, p in enumerate(flat.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(j)))

    def close(self) -> None:
        """removes the autograd hooks (they hold this object, which holds the parameters: a reference cycle)"""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self.active = False
        self.lane_fill = None

    def _make_hook(self, j: int):
        def hook(p):
            # a gradient sink that has claimed this parameter announces it itself, when its kernel is really issued
            # (deferred weight gradients are issued later than autograd visits the leaf)
            if self.held is not None and j in self.held:
                return
            self.notify(j)
        return hook

    def notify(self, j: int) -> None:
        """Gradient of parameter j is complete (called by the autograd hook, or by the gradient sink when the
        backward kernels write the flat buffer directly)."""
        if not self.armed or not self.active:
            return
        # A parameter can be announced twice in one backward: by the gradient sink when its kernel is enqueued and
        # again by autograd's post-accumulate hook (the engine still visits the leaf although the Function returned
        # None for it).  Count each parameter once, or a bucket is launched before its last member is written.
        if j in self.seen:
            return
        self.seen.add(j)
        if self.flat.grad.is_cuda:              # gradients may be produced on several streams
            self.streams.add(torch.cuda.current_stream(self.flat.grad.device))
        b = self.param_bucket[j]
        self.pending[b] -= 1
        if self.pending[b] == 0:
            self._launch(b)

    def _launch(self, b: int) -> None:
        if self.launched[b]:
            return
        lo, hi = self.buckets[b]
        self.launched[b] = True
        if self.flat.grad.is_cuda:
            # The collective must see every producer stream's work, but the compute streams must NOT wait for each
            # other here (that would serialise the weight-gradient side stream with the main backward once per
            # bucket): a dedicated communication stream collects the dependencies and RCCL's stream chains off it.
            dev = self.flat.grad.device
            if self.comm_stream is None:
                self.comm_stream = torch.cuda.Stream(device=dev)
            cs = self.comm_stream
            cs.wait_stream(torch.cuda.current_stream(dev))
            for s in self.streams:
                cs.wait_stream(s)
            with torch.cuda.stream(cs):
                self.works.append(dist.all_reduce(self.flat.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.pg,
                                                  async_op=True))
            return
        self.works.append(dist.all_reduce(self.flat.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.pg,
                                          async_op=True))

    def start(self) -> None:
        self.seen = set()
        self.pending = list(self.bucket_size)
        self.launched = [False] * len(self.buckets)
        self.works = []
        self.armed = True

    def finish(self) -> None:
        """After backward(): the flat gradient holds the SUM over ranks (divide by world in the optimiser)."""
        self.armed = False
        if not self.active:
            return
        for b in range(len(self.buckets)):
            self._launch(b)
        if self.lane_fill is not None:
            # the caller has joined every stream that ran part of the backward with the current one: whatever any
            # recurrence of this step did to the fault word is visible to the fill kernel
            if self.lane is None:
                self.lane = torch.zeros(4, device=self.flat.grad.device, dtype=torch.float32)
            if self.flat.grad.is_cuda:
                if self.comm_stream is None:
                    self.comm_stream = torch.cuda.Stream(device=self.flat.grad.device)
                cs = self.comm_stream
                cs.wait_stream(torch.cuda.current_stream(self.flat.grad.device))
                with torch.cuda.stream(cs):
                    self.lane_fill(self.lane)
                    self.works.append(dist.all_reduce(self.lane, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
            else:
                self.lane_fill(self.lane)
                self.works.append(dist.all_reduce(self.lane, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        for w in self.works:
            w.wait()
        self.works = []
