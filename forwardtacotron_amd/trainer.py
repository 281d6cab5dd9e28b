"""One optimisation step of ForwardTrainer.train_session (trainer/forward_trainer.py:69-99), MI355X-native.

    loss = L1m(mel) + L1m(mel_post) + f_d*L1m(dur) + f_p*L1m(pitch) + f_e*L1m(energy)
    zero_grad -> backward -> clip_grad_norm_(max_norm) -> Adam.step

Forward, losses and backward are the HIP ops of forwardtacotron_amd.ops; clip + Adam run as two fused
kernels over the flat parameter / gradient buffers (no per-tensor loop, no host sync); with a process
group the flat gradient is sum-all-reduced in buckets overlapped with backward (RCCL over xGMI) and the
1/world factor is folded into the clip coefficient.  BatchNorm statistics stay per rank (the reference has
no SyncBN), every rank applies the identical Adam update.

Recurrence faults.  The persistent recurrences poll other workgroups with bounded spins; a poll that runs out leaves
garbage in that step's activations / gradients and sets the device's sticky fault word.  The clip kernel reads the word
ON THE DEVICE: the Adam kernel then leaves parameters and moments untouched, the reported grad_norm is NaN,
`out['rnn_fault']` is 1, and the buffers the forward pass updated in place -- BatchNorm running statistics,
num_batches_tracked, `step` -- are put back from the snapshot taken at the start of the step (parallel.FlatBuffers,
ft_guarded_restore): a faulted step leaves no trace in the model.  Under data parallelism the fault is made GLOBAL before
anything is applied: every rank all-reduces a fault lane with its gradient buckets (parallel.BucketedAllReduce) and every
rank skips alike (`rnn_fault` = 2 on the ranks that only heard of it) -- the faulting rank's garbage gradient has been
summed into everybody's buckets by then.  The host learns about it without a per-step sync: the flag of every step is
copied to pinned memory asynchronously and looked at when a later step (or `check()`) finds the copy complete; with more
than one rank the flag of step s is consumed exactly at the start of step s + 2 on EVERY rank (a wait that is already
over: the host is never two steps ahead of the device), so that all ranks rewind Adam's step count and switch kernels at
the same step.  Policy `on_rnn_fault`: 'raise' (default) raises FtError; 'fallback' clears the word, switches the library
to the per-step recurrence kernels and carries on.
"""
import gc
import os
from typing import Dict, Optional

import torch
import torch.distributed as dist

from . import _lib
from . import hip as H
from . import ops
from .parallel import BucketedAllReduce, FlatBuffers, FlatParams

_CU_LIMITED_STREAMS: Dict[tuple, 'torch.cuda.Stream'] = {}
_WARNED_CUS = False

DEFAULT_TRAIN_CFG = dict(dur_loss_factor=0.1, pitch_loss_factor=0.1, energy_loss_factor=0.1,
                         pitch_zoneout=0.0, energy_zoneout=0.0, clip_grad_norm=1.0)


class TrainStep:
    def __init__(self, model: torch.nn.Module, lr: float, train_cfg: Optional[dict] = None,
                 betas=(0.9, 0.999), eps: float = 1e-8, process_group=None, bucket_bytes: int = 24 << 20,
                 on_rnn_fault: str = 'raise', gc_freeze: Optional[bool] = None):
        """gc_freeze: after the third step, move every object alive in the process out of Python's cyclic GC
        (gc.freeze()).  A full collection walks all modules / parameters / caches -- 30-70 ms of host stall every ~25
        steps with the GPU running dry (profiles/r02_per_step_ms.txt) -- but the freeze is PROCESS-WIDE, so it is the
        application's decision: off unless asked for here or with FT_GC_FREEZE=1 (bench.py and tools/ ask for it);
        close() undoes it."""
        self.model = model
        self.lr = float(lr)
        self.cfg = dict(DEFAULT_TRAIN_CFG)
        if train_cfg:
            self.cfg.update(train_cfg)
        self.betas, self.eps = betas, eps
        self.flat = FlatParams(model)
        dev = self.flat.flat.device
        if dev.type != 'cuda':
            raise _lib.FtError('TrainStep needs the model on an MI355X (HIP) device')
        self.bufs = FlatBuffers(model)
        self.exp_avg = torch.zeros_like(self.flat.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat.flat)
        self.opt_step = 0
        # ft_clip_grad_norm's record: [clip coefficient, grad norm, recurrence-fault flag, 0]
        self.coef = torch.zeros(4, device=dev, dtype=torch.float32)
        if on_rnn_fault not in ('raise', 'fallback'):
            raise ValueError("on_rnn_fault must be 'raise' or 'fallback'")
        self.on_rnn_fault = on_rnn_fault
        self._fault_slots = []          # ring of (pinned host float[1], event, optimizer step it belongs to or None)
        self._fault_next = 0
        self.skipped_steps = 0          # optimizer steps the device skipped because of a recurrence fault
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.reducer = BucketedAllReduce(self.flat, process_group, bucket_bytes,
                                         lane_fill=lambda lane: _lib.call('ft_fault_lane_set', lane.data_ptr(), H._stream()))
        # weight gradients are written straight into the flat buffer, the GEMM-shaped ones on a side stream
        # models with a recurrent trunk: 28 CUs per XCD by default (see _make_wgrad_stream); FT_WGRAD_CUS overrides
        self.wgrad_stream = self._make_wgrad_stream(dev, None if 'FT_WGRAD_CUS' in os.environ
                                                    else (28 if hasattr(model, 'lstm') else 0))
        self.reducer.streams.add(self.wgrad_stream)
        self.sink = ops.GradSink({p.data_ptr(): (i, p.grad) for i, p in enumerate(self.flat.params)},
                                 stream=self.wgrad_stream, on_write=self.reducer.notify)
        self.reducer.held = self.sink.held
        self.sink.reducer_streams = self.reducer.streams     # streams a bucket's all-reduce must wait for
        self.packs: Optional[H.PackCache] = None
        self.main_stream = (torch.cuda.Stream(device=dev, priority=-1)
                            if os.environ.get('FT_MAIN_PRIORITY', '1') == '1' else None)
        self._packs_base = 0
        if gc_freeze is None:
            gc_freeze = os.environ.get('FT_GC_FREEZE', '0') == '1'
        self._gc_frozen = not gc_freeze         # "nothing left to do"
        self._did_freeze = False

    def close(self) -> None:
        """Undoes what this object did to the process: unfreezes the cyclic GC (if it froze it) and breaks the
        parameter -> hook -> reducer -> FlatParams -> parameter cycle so that a dropped TrainStep (and its flat gradient
        and Adam moments on the device) can be collected."""
        if self._did_freeze:
            gc.unfreeze()
            self._did_freeze = False
        self._gc_frozen = True
        self.reducer.close()
        self.sink.on_write = None

    @staticmethod
    def _make_wgrad_stream(dev, n: Optional[int] = None) -> 'torch.cuda.Stream':
        """The weight-gradient side stream, restricted to n CUs of every XCD (ft_stream_create_cu_limited; 0 / 32: an
        ordinary stream).  A weight-gradient GEMM is ONE resident wave of long-running workgroups that fills the register
        file of every CU it may use, so the step's critical stream -- however high its priority -- finds no slot for its
        small dependent kernels until the whole GEMM has drained: in the tail of a ForwardTacotron step the LSTM's four
        weight gradients and the prenet's backward chain of ~150 small kernels ran one after the other in effect.  With 4
        CUs per XCD out of the side stream's reach (and the weight-gradient planner sizing its one-wave grids by the
        stream's slots, ft_stream_slots) the recurrent models gain 0.3 ms per step (24.15 -> 23.85, same box; 24 / 26 /
        30 CUs: 23.85-24.0 / 23.95 / 23.93, 20: 24.3); FastPitch, all of whose weight gradients share the stream with
        nothing recurrent to hide behind, loses (bf16 18.2 -> 19.0) and keeps an ordinary stream."""
        import ctypes
        if n is None:
            n = int(os.environ.get('FT_WGRAD_CUS', '0'))
        if n <= 0 or n >= 32:
            return torch.cuda.Stream(device=dev)
        # the mask layout (bit i = CU i/8 of XCD i%8) is that of a whole MI355X: 8 XCDs x 32 CUs.  On anything else (a
        # DPX / CPX partition, another part) an ordinary stream is used -- unless FT_WGRAD_CUS asked for the mask by name
        cus, is950 = ctypes.c_int(0), ctypes.c_int(0)
        with torch.cuda.device(dev):
            _lib.call('ft_device_info', ctypes.byref(cus), ctypes.byref(is950))
        if (cus.value != 256 or not is950.value) and 'FT_WGRAD_CUS' not in os.environ:
            global _WARNED_CUS
            if not _WARNED_CUS:
                _WARNED_CUS = True
                import warnings
                warnings.warn(f'weight-gradient stream: {cus.value} CUs visible (not a whole MI355X): no CU mask')
            return torch.cuda.Stream(device=dev)
        # one such stream per (device, n) and process, shared by every TrainStep and never destroyed: the caching
        # allocator may still record events on it for tensors that outlive the TrainStep which used it
        key = (torch.device(dev).index or 0, n)
        st = _CU_LIMITED_STREAMS.get(key)
        if st is None:
            with torch.cuda.device(dev):
                h = ctypes.c_void_p()
                _lib.call('ft_stream_create_cu_limited', n, ctypes.byref(h))
            st = _CU_LIMITED_STREAMS[key] = torch.cuda.ExternalStream(h.value, device=dev)
        return st

    @staticmethod
    def _predictors_first(model, B: int) -> bool:
        """Three-stage backward: may the predictors' stage go first?  Yes iff the largest predictor BPTT grid and the
        trunk's first BPTT (the postnet CBHG's GRU) are admitted side by side (ft_rnn_persist.hip, admission: per-XCD demand
        of a backward GRU = H/16 chunks x ceil(groups / 8) over 32 one-workgroup CUs, budget 0.75).  FT_PRED_STAGE_FIRST=0/1
        overrides.  ForwardTacotron bs=32: 0.5 + 0.25 -> first (24.0 -> 23.6, 23.4 -> 23.1 ms on two boxes);
        MultiForwardTacotron (a 256-wide pitch predictor): 0.5 + 0.5 -> not first (measured: 27.4 -> 27.8, bs=64 38.0 -> 38.9)."""
        env = os.environ.get('FT_PRED_STAGE_FIRST')
        if env is not None:
            return env == '1'
        post = getattr(getattr(model, 'postnet', None), 'rnn', None)
        preds = [m.rnn for n, m in model.named_children() if n.endswith('_pred') and hasattr(m, 'rnn')]
        if post is None or not preds or not hasattr(post, 'hidden_size'):
            return False
        # per-XCD shares of the BPTT grids, from the launcher's own occupancy arithmetic (ft_rnn_xcd_fill_pct); without a
        # device to query (host-only tests) the formula in the docstring
        groups = 2 * ((B + 15) // 16)

        def demand(h):
            pct = H._lib.lib().ft_rnn_xcd_fill_pct(3, 1, int(B), 64, int(h)) if torch.cuda.is_available() else -1
            return (h // 16) * ((groups + 7) // 8) / 32.0 * 100.0 if pct < 0 else pct
        return demand(post.hidden_size) + max(demand(r.hidden_size) for r in preds) <= H._lib.lib().ft_rnn_admit_budget_pct()

    def _weight_packs(self) -> H.PackCache:
        """the re-laid-out weight copies of this model (hip.PackCache), rebuilt only if the flat buffer moved"""
        if self.packs is None or self._packs_base != self.flat.flat.data_ptr():
            in_bank, banks = set(), []
            for mod in self.model.modules():
                if hasattr(mod, 'conv1d_bank'):
                    ws = [m.conv.weight for m in mod.conv1d_bank]
                    banks.append(ws)
                    in_bank.update(id(w) for w in ws)
            emb = {id(p) for m in self.model.modules() if isinstance(m, torch.nn.Embedding) for p in m.parameters()}
            ps = [p for p in self.flat.params if p.dtype == torch.float32 and id(p) not in emb]
            highways = [(m.W1.weight, m.W2.weight) for m in self.model.modules()
                        if hasattr(m, 'W1') and hasattr(m, 'W2') and isinstance(m.W1, torch.nn.Linear)]
            self.packs = H.PackCache(mats=[p for p in ps if p.dim() == 2],
                                     convs=[p for p in ps if p.dim() == 3 and id(p) not in in_bank],
                                     banks=banks, device=self.flat.flat.device, highways=highways)
            self._packs_base = self.flat.flat.data_ptr()
        return self.packs

    # -- state for checkpoints: same content as torch.optim.Adam's (exp_avg / exp_avg_sq / step), flat
    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {'exp_avg': self.exp_avg, 'exp_avg_sq': self.exp_avg_sq,
                'step': torch.tensor(self.opt_step), 'lr': torch.tensor(self.lr)}

    def side_losses(self, pred: Dict[str, torch.Tensor], batch: Dict[str, torch.Tensor], pitch_target, energy_target,
                    dur_target=None):
        """the predictors' loss terms (forward_trainer.py:86-93) and their weighted sum"""
        c = self.cfg
        dur_t = batch['dur'] if dur_target is None else dur_target
        dl = ops.masked_l1(pred['dur'].unsqueeze(1), dur_t.unsqueeze(1), batch['x_len'])
        pl = ops.masked_l1(pred['pitch'], pitch_target.unsqueeze(1), batch['x_len'])
        el = ops.masked_l1(pred['energy'], energy_target.unsqueeze(1), batch['x_len'])
        side = c['dur_loss_factor'] * dl + c['pitch_loss_factor'] * pl + c['energy_loss_factor'] * el
        out = {'dur': dl, 'pitch': pl, 'energy': el}
        if 'pitch_cond' in pred:        # multispeaker: CrossEntropyLoss(ignore_index=0), multi_forward_trainer.py:34,88
            ce = ops.cross_entropy(pred['pitch_cond'], batch['pitch_cond'], 0)
            side = side + c.get('pitch_cond_loss_factor', 0.1) * ce
            out['pitch_cond'] = ce
        return out, side

    def losses(self, pred: Dict[str, torch.Tensor], batch: Dict[str, torch.Tensor], pitch_target, energy_target):
        """forward_trainer.py:83-93"""
        m1 = ops.masked_l1(pred['mel'], batch['mel'], batch['mel_len'])
        m2 = ops.masked_l1(pred['mel_post'], batch['mel'], batch['mel_len'])
        out, side = self.side_losses(pred, batch, pitch_target, energy_target)
        out = dict(mel=m1, mel_post=m2, **out)
        out['loss'] = m1 + m2 + side
        # the same total as two roots (see _step: the predictors' backward is a stage of its own)
        out['_roots'] = (side, m1 + m2)
        return out

    def step(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """batch: device tensors with the ForwardCollator layout (utils/dataset.py:239-263).  Returns the
        loss terms and the pre-clip gradient norm as device scalars (no host sync in here except the
        LengthRegulator's output-size read, which the reference has too)."""
        # a fault of an EARLIER step whose flag has reached the host by now; with several ranks: exactly the flags of the
        # steps up to two back, on every rank (see the module docstring)
        self._poll_faults(wait=self.world > 1, upto=self.opt_step - 1 if self.world > 1 else None)
        if self.main_stream is None:
            return self._step(batch)
        # the step's critical path runs on a HIGH-priority stream: its bandwidth-bound kernels (BatchNorm statistics,
        # pooling gradients) otherwise queue behind the side stream's weight-gradient GEMMs, whose workgroups fill the
        # register files of every CU
        cur = torch.cuda.current_stream()
        self.main_stream.wait_stream(cur)
        with torch.cuda.stream(self.main_stream):
            out = self._step(batch)
        cur.wait_stream(self.main_stream)
        for v in out.values():
            v.record_stream(cur)
        return out

    def _step(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        model = self.model
        model.train()
        if not self.flat.attached():
            self.flat.attach()
        if not self.bufs.attached():
            self.bufs.attach()
        self.bufs.snapshot()                # what a faulted step is rolled back to (optimizer_step)
        c = self.cfg
        pitch_target = batch['pitch'].detach().clone()
        energy_target = batch['energy'].detach().clone()
        if c['pitch_zoneout'] > 0 or c['energy_zoneout'] > 0:
            batch = dict(batch)
            dev = batch['x'].device
            pm = (torch.rand(batch['x'].size()) > c['pitch_zoneout']).to(dev).float()
            em = (torch.rand(batch['x'].size()) > c['energy_zoneout']).to(dev).float()
            batch['pitch'] = batch['pitch'] * pm
            batch['energy'] = batch['energy'] * em
        packs = self._weight_packs()
        packs.refresh()                     # every conv pack / weight transpose of this step, one launch
        H.pack_cache = packs
        old_precision = H.set_gemm_precision(getattr(model, 'matmul_dtype', 'fp32'))    # forward AND backward
        independent = bool(getattr(model, 'independent_predictors', False))
        # EARLY predictor backward (FT_PRED_BWD_EARLY=1, off by default): the predictors' loss terms depend on nothing but
        # the predictors' own outputs and the batch's targets (forward_trainer.py:86-93), so their whole backward can be
        # issued on the predictors' stream right behind their forward, INSIDE the model's forward (model.predictor_hook),
        # beside the trunk's forward recurrences instead of in the tail of the step.  Measured:
        # the tail does not get shorter (it is bound by the main stream's own chain and the LSTM's weight gradients, the
        # predictors' stage ran in their shadow) while the LSTM's forward recurrence stretches 4.1 -> 5.5 ms beside the
        # predictors' BPTT kernels: 24.6 -> 25.8 ms.  Same gradients bit for bit either way (tests run both).
        early = independent and os.environ.get('FT_PRED_BWD_EARLY', '0') == '1'
        # staged backward (see below): only this trainer asks the model to cut its graph below the LSTM
        staged = (not early and independent and hasattr(model, 'lstm')
                  and os.environ.get('FT_STAGED_BACKWARD', '1') == '1')
        model.stage_backward = staged
        model._cut = None
        side_terms: Dict[str, torch.Tensor] = {}

        def arm_sink():
            self.sink.inline_rows = int(getattr(model, 'wgrad_inline_rows', 0))
            # models with recurrences queue their side-stream weight gradients and issue them beside the next BPTT kernel
            self.sink.defer = bool(getattr(model, 'wgrad_defer', False)) and os.environ.get('FT_WGRAD_DEFER', '1') == '1'

        def predictor_hook(p: Dict[str, torch.Tensor]) -> None:
            # called by the model on the predictors' stream, right behind their forward
            arm_sink()
            # the LengthRegulator clamps batch['dur'] in place (on the main stream, possibly at this very moment) and the
            # reference's loss sees the clamped tensor (forward_trainer.py:79-86): clamp a copy, either read gives the same
            terms, root = self.side_losses(p, batch, pitch_target, energy_target, dur_target=batch['dur'].clamp(min=0.0))
            root.backward()                 # ends by joining the CALLING stream (this one) with the streams it used
            self.sink.used.add(torch.cuda.current_stream())
            side_terms.update({k: v.detach() for k, v in terms.items()})
            side_terms['_root'] = root.detach()

        try:
            self.flat.zero_grad()
            self.reducer.start()
            self.sink.begin_step()
            if early:
                model.predictor_hook = predictor_hook
                ops.set_grad_sink(self.sink)
            try:
                pred = model(batch)
            finally:
                model.stage_backward = False
                model.predictor_hook = None
            arm_sink()
            ops.set_grad_sink(self.sink)
            for v in side_terms.values():       # allocated on the predictors' stream, read from here on
                v.record_stream(torch.cuda.current_stream())
            if early and '_root' in side_terms:
                m1 = ops.masked_l1(pred['mel'], batch['mel'], batch['mel_len'])
                m2 = ops.masked_l1(pred['mel_post'], batch['mel'], batch['mel_len'])
                (m1 + m2).backward()
                ops.flush_deferred()
                L = dict(mel=m1, mel_post=m2, **{k: v for k, v in side_terms.items() if k != '_root'})
                L['loss'] = m1.detach() + m2.detach() + side_terms['_root']
            else:
                L = self.losses(pred, batch, pitch_target, energy_target)
                # The predictor branches share nothing with the trunk (each has its own embedding; the trunk is fed the
                # batch's pitch / energy / pitch_cond, forward_tacotron.py:129-159) and run on their own side stream.
                # One backward over the summed loss issues their nodes LAST (autograd orders ready nodes by creation
                # order, the predictors were created first): 1.9 ms of predictor BPTT + conv backward at the end of the
                # step with the main stream idle.  Issued FIRST they collide with the trunk's recurrences instead (a
                # persistent LSTM BPTT fills its XCD slots: it would wait for every queued predictor BPTT, +1 ms).  So
                # the backward runs in three stages: postnet .. LSTM (down to the cut below the LSTM), then the
                # predictors -- their BPTT kernels queue behind the LSTM's and run beside the prenet's GEMMs -- then
                # LR .. prenet, whose GRU fits next to a predictor's.
                side_root, main_root = L.pop('_roots')
                cut = getattr(model, '_cut', None)
                model._cut = None
                first = staged and cut is not None and self._predictors_first(model, int(batch['x'].shape[0]))
                self.sink.late_ok = first and os.environ.get('FT_WGRAD_LATE', '1') == '1'
                if first:
                    # The predictors' stage issued FIRST: it then runs beside the postnet GRU's BPTT and the postnet's conv
                    # backward -- the first 4 ms of the backward, of which 2 ms are a recurrence that leaves most of the chip
                    # idle -- instead of in the tail of the step, where it was the last stream to finish (the LSTM's BPTT,
                    # which fills its XCD slots, waits on the device for the predictors' BPTT kernels: they are long done).
                    # Only where the predictors' BPTT grids fit BESIDE the postnet GRU's (_predictors_first): otherwise they
                    # queue behind it and the stage delays the trunk (multispeaker: 38.0 -> 38.9 ms).
                    here = torch.cuda.current_stream()
                    pstream = model._side_stream(side_root.device)
                    pstream.wait_stream(here)
                    with torch.cuda.stream(pstream):
                        side_root.backward()
                    side_root.record_stream(pstream)
                    self.sink.used.add(pstream)
                    main_root.backward()
                    cut[0].backward(cut[1].grad)
                elif staged and cut is not None:
                    main_root.backward()
                    # backward() ends by making the CALLING stream wait for every stream it ran nodes on: called from
                    # the main stream the predictors' stage would simply be inserted into the critical path.  It is
                    # issued from the predictors' own stream; the step's final join picks that stream up.
                    here = torch.cuda.current_stream()
                    pstream = model._side_stream(side_root.device)
                    pstream.wait_stream(here)
                    with torch.cuda.stream(pstream):
                        side_root.backward()
                    side_root.record_stream(pstream)
                    self.sink.used.add(pstream)
                    cut[0].backward(cut[1].grad)
                else:
                    L['loss'].backward()
                ops.flush_deferred()
        finally:
            ops.set_grad_sink(None)
            H.pack_cache = None
            packs.release()
            H.set_gemm_precision(old_precision)
        cur = torch.cuda.current_stream()
        cur.wait_stream(self.wgrad_stream)
        for st in self.sink.used:               # e.g. the predictors' side stream: its backward wrote gradients too
            if st != cur and st != self.wgrad_stream:
                cur.wait_stream(st)
        self.sink.keep.clear()                  # joined: nothing reads the side launches' operands any more
        self.reducer.finish()
        self.optimizer_step()
        out = {k: v.detach() for k, v in L.items()}
        out['grad_norm'] = self.coef[1]         # NaN if a recurrence faulted (the update was skipped on the device)
        out['rnn_fault'] = self.coef[2]
        self._post_fault_flag()
        if not self._gc_frozen and self.opt_step >= 3:
            # Everything long-lived exists by now (parameters, modules, weight packs, workspaces, streams).  A full
            # collection of Python's cyclic GC walks all of it: 30-70 ms of host stall every ~20-30 steps, during which
            # the GPU runs dry (one such pause inside 20 timed steps = +3.5 ms/step).  Moved to the permanent generation
            # those objects are no longer scanned; the per-step garbage (autograd graph cycles) stays collectable and
            # cheap.  gc.unfreeze() undoes it (FT_GC_FREEZE=0: never frozen).
            gc.collect()
            gc.freeze()
            self._gc_frozen = True
            self._did_freeze = True
        return out

    # -- recurrence-fault surfacing (no per-step host sync) ------------------------------------------------
    def _post_fault_flag(self) -> None:
        """async copy of this step's fault flag into a pinned slot; looked at by a later step / check()"""
        if len(self._fault_slots) < 8:
            self._fault_slots.append([torch.zeros(1, dtype=torch.float32).pin_memory(), torch.cuda.Event(), None])
        slot = self._fault_slots[self._fault_next % len(self._fault_slots)]
        self._fault_next += 1
        if slot[2] is not None:             # ring wrapped onto a copy nobody has looked at yet: look now
            slot[1].synchronize()
            self._consume(slot)
        slot[0].copy_(self.coef[2:3], non_blocking=True)
        slot[1].record()
        slot[2] = self.opt_step

    def _consume(self, slot) -> None:
        step_no, flag = slot[2], float(slot[0][0])
        slot[2] = None
        if flag != 0.0:
            self._handle_fault(step_no, remote=flag == 2.0)

    def _poll_faults(self, wait: bool, upto: Optional[int] = None) -> None:
        for slot in sorted((s for s in self._fault_slots if s[2] is not None), key=lambda s: s[2]):
            if slot[2] is None:             # an earlier slot's fault handling already dealt with everything pending
                continue
            if upto is not None and slot[2] > upto:
                continue
            if wait:
                slot[1].synchronize()
            elif not slot[1].query():
                continue
            self._consume(slot)

    def _handle_fault(self, first_bad_step: int, remote: bool = False) -> None:
        # the fault word is sticky: every optimizer step from `first_bad_step` on was skipped on the device
        torch.cuda.synchronize()
        self.skipped_steps += self.opt_step - (first_bad_step - 1)
        self.opt_step = first_bad_step - 1          # Adam's bias correction continues where the last real update was
        for slot in self._fault_slots:
            slot[2] = None
        try:
            H.check_rnn_status(clear=True)
        except _lib.FtError:
            pass
        where = 'on ANOTHER rank ' if remote else ''
        msg = (f'persistent recurrence timed out {where}in optimizer step {first_bad_step}: that update (and every later '
               f'one) was skipped on the device{" of every rank" if self.world > 1 else ""}; parameters, Adam moments, '
               f'BatchNorm running statistics and step counters are intact')
        if self.on_rnn_fault == 'fallback':
            _lib.query('ft_rnn_set_persistent', 0)
            import warnings
            warnings.warn(msg + '; continuing with the per-step recurrence kernels')
            return
        raise _lib.FtError(msg + "; construct TrainStep(on_rnn_fault='fallback') to continue on the per-step kernels")

    def check(self) -> None:
        """Waits for every step issued so far and raises (or falls back) if one of them hit a recurrence fault."""
        self._poll_faults(wait=True)

    def optimizer_step(self) -> None:
        f = self.flat
        ws = H.workspace(_lib.query('ft_grad_norm_workspace'), f.flat.device)
        max_norm = self.cfg.get('clip_grad_norm') or 0.0
        lane = self.reducer.lane            # the all-reduced fault lane (None: no process group)
        _lib.call('ft_clip_grad_norm', f.grad.data_ptr(), f.total, float(max_norm), 1.0 / self.world,
                  lane.data_ptr() if lane is not None else None, self.coef.data_ptr(), ws.data_ptr(), ws.numel(),
                  H._stream())
        b = self.bufs                       # a faulted step leaves no trace in the forward-updated buffers either
        if b.stats.numel():
            _lib.call('ft_guarded_restore', b.stats.data_ptr(), b.stats_snap.data_ptr(), b.stats.numel(),
                      self.coef.data_ptr(), H._stream())
        if b.counts.numel():
            _lib.call('ft_guarded_restore', b.counts.data_ptr(), b.counts_snap.data_ptr(), 2 * b.counts.numel(),
                      self.coef.data_ptr(), H._stream())
        self.opt_step += 1
        _lib.call('ft_adam_step', f.flat.data_ptr(), f.grad.data_ptr(), self.exp_avg.data_ptr(),
                  self.exp_avg_sq.data_ptr(), f.total, self.lr, self.betas[0], self.betas[1], self.eps,
                  self.opt_step, self.coef.data_ptr(), H._stream())
