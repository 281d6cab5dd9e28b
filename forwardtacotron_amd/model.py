"""MI355X-native ForwardTacotron: same constructor, batch-dict forward()/generate() and state_dict as
the reference (models/forward_tacotron.py:14-254, models/common_layers.py:12-124) -- 322 state_dict
entries with identical names / shapes / dtypes, identical default initialisation under the same seed --
but every forward and backward computation is a hand-written gfx950 kernel behind the C ABI.

torch.nn.{Embedding,Conv1d,BatchNorm1d,Linear} objects appear below ONLY as parameter containers (they
give the reference's exact names and default init); their forward() is never called.  There is no
CPU / eager fallback: the model refuses to run if its tensors are not on a HIP device.
"""
import math
import os
from pathlib import Path
from typing import Any, Callable, Dict, List, Optional, Union

import torch
import torch.nn as nn

from . import hip as H
from . import ops

PAD_VALUE = -11.5129
NUM_CHARS_DEFAULT = 135      # len(utils.text.symbols.phonemes), utils/text/symbols.py:21-23


_seed_state = {'torch_seed': None, 'base': 0, 'n': 0}


def _seed() -> int:
    """Seed of one dropout site: host arithmetic only (no device sync, no tensor op -- a step draws ~170 of them).
    The stream is re-based from torch's host RNG whenever torch.manual_seed() installs a different seed."""
    st = _seed_state
    s = torch.initial_seed()
    if st['torch_seed'] != s:
        st['torch_seed'] = s
        st['base'] = int(torch.randint(0, 2 ** 62, (1,)).item())
        st['n'] = 0
    st['n'] += 1
    return (st['base'] + 0x9E3779B97F4A7C15 * st['n']) & ((1 << 62) - 1)


def _side_priority() -> int:
    import os
    return -1 if os.environ.get('FT_PRED_PRIORITY', '1') == '1' else 0


def _dropout(x: torch.Tensor, p: float, training: bool) -> torch.Tensor:
    if not training or p <= 0.0:
        return x
    return ops.DropoutFn.apply(x, p, _seed())


class _RNNParams(nn.Module):
    """Parameter container with nn.GRU / nn.LSTM (1 layer, bidirectional) names and init
    (uniform(-1/sqrt(H), 1/sqrt(H)) in registration order, like torch's RNNBase.reset_parameters)."""

    def __init__(self, input_size: int, hidden_size: int, gates: int) -> None:
        super().__init__()
        self.input_size, self.hidden_size, self.gates = input_size, hidden_size, gates
        for sfx in ('', '_reverse'):
            self.register_parameter('weight_ih_l0' + sfx, nn.Parameter(torch.empty(gates * hidden_size, input_size)))
            self.register_parameter('weight_hh_l0' + sfx, nn.Parameter(torch.empty(gates * hidden_size, hidden_size)))
            self.register_parameter('bias_ih_l0' + sfx, nn.Parameter(torch.empty(gates * hidden_size)))
            self.register_parameter('bias_hh_l0' + sfx, nn.Parameter(torch.empty(gates * hidden_size)))
        stdv = 1.0 / math.sqrt(hidden_size) if hidden_size > 0 else 0
        for w in self.parameters():
            nn.init.uniform_(w, -stdv, stdv)

    def weights(self) -> List[torch.Tensor]:
        return [self.weight_ih_l0, self.weight_hh_l0, self.bias_ih_l0, self.bias_hh_l0,
                self.weight_ih_l0_reverse, self.weight_hh_l0_reverse, self.bias_ih_l0_reverse,
                self.bias_hh_l0_reverse]


class GRU(_RNNParams):
    """nn.GRU(in, H, batch_first=True, bidirectional=True) replacement (common_layers.py:89)."""

    def __init__(self, input_size: int, hidden_size: int) -> None:
        super().__init__(input_size, hidden_size, 3)

    def forward(self, x: torch.Tensor, time_major_out: bool = False) -> torch.Tensor:
        """x [B,T,in] -> [B,T,2H]; time_major_out=True returns the recurrence's native [T,B,2H] layout
        (what the fused consumers inside this package read directly)."""
        y = ops.BiGRUFn.apply(x, *self.weights())
        return y if time_major_out else ops.BTTransposeFn.apply(y, False)


class LSTM(_RNNParams):
    """nn.LSTM(in, H, batch_first=True, bidirectional=True) replacement with the pack/unpack semantics of
    forward_tacotron.py:147-152 built in (lens + padding_value)."""

    def __init__(self, input_size: int, hidden_size: int) -> None:
        super().__init__(input_size, hidden_size, 4)

    def forward(self, x: torch.Tensor, lens: Optional[torch.Tensor], pad_value: float) -> torch.Tensor:
        return ops.BiLSTMFn.apply(x, lens, pad_value, *self.weights())

    def forward_regulated(self, x: torch.Tensor, dur: torch.Tensor, lens: Optional[torch.Tensor],
                          pad_value: float) -> torch.Tensor:
        """self(LengthRegulator()(x, dur, lens), lens, pad_value) as one node, with the input projection formed per
        token instead of per frame (ops.LRBiLSTMFn); x is the token-level input [B,Tx,I]"""
        if not dur.is_contiguous() or dur.dtype != torch.float32:
            raise H._lib.FtError('LengthRegulator: dur must be contiguous fp32 (it is clamped in place)')
        return ops.LRBiLSTMFn.apply(x, dur, lens, pad_value, *self.weights())


def regulate_and_decode(model, x: torch.Tensor, dur: torch.Tensor, mel_lens: Optional[torch.Tensor]) -> torch.Tensor:
    """LengthRegulator + decoder LSTM of the ForwardTacotron variants (forward_tacotron.py:145-152), and the place where
    trainer.TrainStep's staged backward cuts the graph.  FT_LR_LSTM_FUSED=0 keeps the two as separate nodes."""
    staged = model.training and torch.is_grad_enabled() and getattr(model, 'stage_backward', False)
    fused = os.environ.get('FT_LR_LSTM_FUSED', '1') != '0'
    if not fused:
        x = model.lr(x, dur, mel_lens)      # at max(mel_lens) frames, the length pad_packed_sequence returns (:147-152)
    if staged:
        # trainer.TrainStep runs the backward in three stages (postnet .. LSTM | predictors | LR .. prenet): the graph is
        # cut here, below the (regulated) LSTM, and the trainer feeds the cut's gradient into the lower part itself
        cut = x.detach().requires_grad_(True)
        model._cut = (x, cut)
        x = cut
    if fused:
        return model.lstm.forward_regulated(x, dur, mel_lens, model.padding_value)
    return model.lstm(x, mel_lens, model.padding_value)


class LengthRegulator(nn.Module):
    """common_layers.py:12-24"""

    def forward(self, x: torch.Tensor, dur: torch.Tensor, pack_lens: Optional[torch.Tensor] = None) -> torch.Tensor:
        """pack_lens: lengths the result will be packed with (see ops.LengthRegulateFn); None = reference signature"""
        if not dur.is_contiguous() or dur.dtype != torch.float32:
            raise H._lib.FtError('LengthRegulator: dur must be contiguous fp32 (it is clamped in place)')
        return ops.LengthRegulateFn.apply(x, dur, pack_lens)


class HighwayNetwork(nn.Module):
    """common_layers.py:27-40"""

    def __init__(self, size: int) -> None:
        super().__init__()
        self.W1 = nn.Linear(size, size)
        self.W2 = nn.Linear(size, size)
        self.W1.bias.data.fill_(0.)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.HighwayFn.apply(x, self.W1.weight, self.W1.bias, self.W2.weight, self.W2.bias)


class BatchNormConv(nn.Module):
    """common_layers.py:43-57 on channels-last input; conv -> ReLU -> BatchNorm (+ residual)."""

    def __init__(self, in_channels: int, out_channels: int, kernel: int, relu=True) -> None:
        super().__init__()
        self.conv = nn.Conv1d(in_channels, out_channels, kernel, stride=1, padding=kernel // 2, bias=False)
        self.bnorm = nn.BatchNorm1d(out_channels)
        self.relu = relu

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        bn = self.bnorm
        if self.training:
            return ops.BatchNormConvFn.apply(x, self.conv.weight, bn.weight, bn.bias, residual, bn.running_mean,
                                             bn.running_var, self.relu is True)
        # eval: BatchNorm folds into the conv epilogue (scale/shift after the ReLU)
        _eval_needs_no_grad(x, self.conv.weight)
        scale, shift = H.bn_fold_eval(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
        wp = H.conv_pack_weight(self.conv.weight)
        acc = _c_clone(residual) if residual is not None else None
        return H.conv1d_fwd(x.contiguous(), wp, relu=self.relu is True, Tout=x.shape[1], scale=scale, shift=shift,
                            accumulate_into=acc)


def _eval_needs_no_grad(x: torch.Tensor, w: torch.Tensor) -> None:
    """The eval-mode BatchNormConv / conv-bank path is raw kernel launches (BatchNorm folded into the convolution's
    epilogue): it records no autograd graph.  The reference's eval forward IS differentiable; rather than hand back
    tensors that silently carry no gradient, refuse the combination (the reference's own eval call sites --
    evaluate(), generate(), create_gta_features -- all run under torch.no_grad())."""
    if torch.is_grad_enabled() and (x.requires_grad or w.requires_grad):
        raise H._lib.FtError('eval-mode forward is not differentiable here (BatchNorm is folded into the convolution '
                             'epilogue): call it under torch.no_grad(), or use model.train() for gradients')


def _c_clone(t: torch.Tensor) -> torch.Tensor:
    return t.contiguous().clone()


class CBHG(nn.Module):
    """common_layers.py:60-124 on channels-last tensors: x [B,T,in] -> [B,T,2*channels]."""

    def __init__(self, K: int, in_channels: int, channels: int, proj_channels: list, num_highways: int,
                 dropout: float = 0.5) -> None:
        super().__init__()
        self.dropout = dropout
        self.bank_kernels = [i for i in range(1, K + 1)]
        self.conv1d_bank = nn.ModuleList()
        for k in self.bank_kernels:
            self.conv1d_bank.append(BatchNormConv(in_channels, channels, k))
        self.conv_project1 = BatchNormConv(len(self.bank_kernels) * channels, proj_channels[0], 3)
        self.conv_project2 = BatchNormConv(proj_channels[0], proj_channels[1], 3, relu=False)
        self.pre_highway = nn.Linear(proj_channels[-1], channels, bias=False)
        self.highways = nn.ModuleList()
        for _ in range(num_highways):
            self.highways.append(HighwayNetwork(channels))
        self.rnn = GRU(channels, channels)
        self._flat = None

    # -- the K BatchNorms of the bank share flat storage so one kernel normalises the whole [B,T,K*C] buffer
    def _bank_flat(self):
        """Returns flat [K*C] aliases (gamma, beta, running_mean, running_var) of the members' tensors.
        Members already adjacent in memory (e.g. placed so by parallel.FlatParams, or by an earlier call)
        are aliased as they are; otherwise they are re-homed into one fresh buffer.  load_state_dict copies
        in place and keeps the aliasing; .to()/.cuda() break it and are repaired here."""
        bns = [m.bnorm for m in self.conv1d_bank]
        K, C = len(bns), bns[0].weight.numel()
        out = []
        for name in ('weight', 'bias', 'running_mean', 'running_var'):
            ts = [getattr(b, name) for b in bns]
            base = ts[0].data_ptr()
            adjacent = all(ts[i].data_ptr() == base + 4 * i * C for i in range(1, K)) and \
                ts[0].untyped_storage().nbytes() - 4 * ts[0].storage_offset() >= 4 * K * C
            if not adjacent:
                flat = torch.cat([t.detach().reshape(-1) for t in ts]).contiguous()
                for i, b in enumerate(bns):
                    view = flat[i * C:(i + 1) * C]
                    if name in ('weight', 'bias'):
                        getattr(b, name).data = view
                    else:
                        b._buffers[name] = view
                ts = [getattr(b, name) for b in bns]
            out.append(ts[0].detach().as_strided((K * C,), (1,)))
        return out

    def forward(self, x: torch.Tensor, time_major_out: bool = False) -> torch.Tensor:
        K = len(self.bank_kernels)
        C = self.conv1d_bank[0].conv.weight.shape[0]
        gamma, beta, rm, rv = self._bank_flat()
        if self.training:
            ws = [m.conv.weight for m in self.conv1d_bank]
            gs = [m.bnorm.weight for m in self.conv1d_bank]
            bs = [m.bnorm.bias for m in self.conv1d_bank]
            y = ops.ConvBankFn.apply(x, K, gamma, beta, rm, rv, *ws, *gs, *bs)
        else:
            B, T, Cin = x.shape
            _eval_needs_no_grad(x, self.conv1d_bank[0].conv.weight)
            scale, shift = H.bn_fold_eval(gamma, beta, rm, rv, self.conv1d_bank[0].bnorm.eps)
            wp_all = torch.cat([H.conv_pack_weight(m.conv.weight).reshape(-1) for m in self.conv1d_bank])
            y = H.maxpool2_fwd(H.conv_bank_fwd(x, wp_all, K, C, relu=True, Tout=T, scale=scale, shift=shift))
        y = _dropout(y, self.dropout, self.training)
        y = self.conv_project1(y)
        y = _dropout(y, self.dropout, self.training)
        y = self.conv_project2(y, residual=x)
        y = ops.LinearFn.apply(y, self.pre_highway.weight, None)
        y = ops.highway_stack(y, list(self.highways))     # gates inside the GEMM epilogues (width % 32 == 0)
        return self.rnn(y, time_major_out=time_major_out)


class SeriesPredictor(nn.Module):
    """forward_tacotron.py:14-39 ; returns [B,T,1]."""

    def __init__(self, num_chars, emb_dim=64, conv_dims=256, rnn_dims=64, dropout=0.5):
        super().__init__()
        self.embedding = nn.Embedding(num_chars, emb_dim)
        self.convs = nn.ModuleList([
            BatchNormConv(emb_dim, conv_dims, 5, relu=True),
            BatchNormConv(conv_dims, conv_dims, 5, relu=True),
            BatchNormConv(conv_dims, conv_dims, 5, relu=True),
        ])
        self.rnn = GRU(conv_dims, rnn_dims)
        self.lin = nn.Linear(2 * rnn_dims, 1)
        self.dropout = dropout

    def forward(self, x: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
        x = ops.EmbeddingFn.apply(x, self.embedding.weight)
        for conv in self.convs:
            x = conv(x)
            x = _dropout(x, self.dropout, self.training)
        B = x.shape[0]
        x = self.rnn(x, time_major_out=True)
        x = ops.LinearFn.apply(x, self.lin.weight, self.lin.bias, B)       # [T,B,2H] -> [B,T,1]
        if alpha != 1.0:
            x = ops.ScaleFn.apply(x, 1.0 / alpha)
        return x


class ForwardTacotron(nn.Module):
    """Drop-in for models/forward_tacotron.py:42-254."""

    def __init__(self,
                 embed_dims: int, series_embed_dims: int, num_chars: int,
                 durpred_conv_dims: int, durpred_rnn_dims: int, durpred_dropout: float,
                 pitch_conv_dims: int, pitch_rnn_dims: int, pitch_dropout: float, pitch_strength: float,
                 energy_conv_dims: int, energy_rnn_dims: int, energy_dropout: float, energy_strength: float,
                 rnn_dims: int, prenet_dims: int, prenet_k: int, postnet_num_highways: int,
                 prenet_dropout: float, postnet_dims: int, postnet_k: int, prenet_num_highways: int,
                 postnet_dropout: float, n_mels: int, padding_value=PAD_VALUE):
        super().__init__()
        self._ctor_kwargs = {k: v for k, v in locals().items() if k not in ('self', '__class__')}
        self.rnn_dims = rnn_dims
        self.padding_value = padding_value
        self.embedding = nn.Embedding(num_chars, embed_dims)
        self.lr = LengthRegulator()
        # predictor branches share no graph node with the trunk in training (trainer.TrainStep may run their backward as a
        # stage of its own)
        self.independent_predictors = True
        self.dur_pred = SeriesPredictor(num_chars=num_chars, emb_dim=series_embed_dims,
                                        conv_dims=durpred_conv_dims, rnn_dims=durpred_rnn_dims,
                                        dropout=durpred_dropout)
        self.pitch_pred = SeriesPredictor(num_chars=num_chars, emb_dim=series_embed_dims,
                                          conv_dims=pitch_conv_dims, rnn_dims=pitch_rnn_dims,
                                          dropout=pitch_dropout)
        self.energy_pred = SeriesPredictor(num_chars=num_chars, emb_dim=series_embed_dims,
                                           conv_dims=energy_conv_dims, rnn_dims=energy_rnn_dims,
                                           dropout=energy_dropout)
        self.prenet = CBHG(K=prenet_k, in_channels=embed_dims, channels=prenet_dims,
                           proj_channels=[prenet_dims, embed_dims], num_highways=prenet_num_highways,
                           dropout=prenet_dropout)
        self.lstm = LSTM(2 * prenet_dims, rnn_dims)
        self.lin = nn.Linear(2 * rnn_dims, n_mels)
        self.register_buffer('step', torch.zeros(1, dtype=torch.long))
        self.postnet = CBHG(K=postnet_k, in_channels=n_mels, channels=postnet_dims,
                            proj_channels=[postnet_dims, n_mels], num_highways=postnet_num_highways,
                            dropout=postnet_dropout)
        self.post_proj = nn.Linear(2 * postnet_dims, n_mels, bias=False)
        self.pitch_strength = pitch_strength
        self.energy_strength = energy_strength
        self.pitch_proj = nn.Conv1d(1, 2 * prenet_dims, kernel_size=3, padding=1)
        self.energy_proj = nn.Conv1d(1, 2 * prenet_dims, kernel_size=3, padding=1)
        self._nbt_flat = None

    def __repr__(self):
        num_params = sum(p.numel() for p in self.parameters())
        return f'ForwardTacotron, num params: {num_params}'

    # ------------------------------------------------------------------------------------------------
    def _require_device(self, t: torch.Tensor) -> None:
        if not t.is_cuda or not self.embedding.weight.is_cuda:
            raise H._lib.FtError('forwardtacotron_amd.ForwardTacotron runs on an MI355X (HIP) device only: '
                                 'move the model and the batch with .cuda(); there is no CPU fallback')

    def _bump_batchnorm_counters(self) -> None:
        """num_batches_tracked += 1 for every BatchNorm1d, as one op on shared int64 storage."""
        bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm1d)]
        f = self._nbt_flat
        ok = f is not None and f.device == bns[0].num_batches_tracked.device
        if ok:
            for i in (0, len(bns) - 1):
                ok = ok and bns[i].num_batches_tracked.data_ptr() == f.data_ptr() + 8 * i
        if not ok:
            f = torch.stack([b.num_batches_tracked.detach().reshape(()) for b in bns]).contiguous()
            for i, b in enumerate(bns):
                b._buffers['num_batches_tracked'] = f[i]
            self._nbt_flat = f
        f += 1

    def _trunk(self, x: torch.Tensor, dur, pitch, energy, mel_lens: Optional[torch.Tensor], late_inputs=None):
        """late_inputs (inference): a callable that delivers (dur, pitch, energy) once the prenet has been enqueued -- the
        predictors then run on a side stream beside embedding + prenet CBHG instead of in front of them"""
        B = x.shape[0]
        x = ops.EmbeddingFn.apply(x, self.embedding.weight)
        x = self.prenet(x, time_major_out=True)                             # [Tx,B,2P] (recurrence layout)
        if late_inputs is not None:
            dur, pitch, energy = late_inputs()
        x = ops.CondAddFn.apply(x, pitch, energy, self.pitch_proj.weight, self.pitch_proj.bias,
                                self.energy_proj.weight, self.energy_proj.bias, self.pitch_strength,
                                self.energy_strength, True)                 # -> [B,Tx,2P]
        x = regulate_and_decode(self, x, dur, mel_lens)
        mel = ops.LinearFn.apply(x, self.lin.weight, self.lin.bias)        # [B,T,n_mels]
        post = self.postnet(mel, time_major_out=True)                       # [T,B,2Q]
        post = ops.LinearFn.apply(post, self.post_proj.weight, None, B)     # -> [B,T,n_mels]
        return mel, post

    def forward(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        x = batch['x']
        mel = batch['mel']
        dur = batch['dur']
        mel_lens = batch['mel_len']
        self._require_device(x)
        pitch = batch['pitch']               # [B,Tx]  (the reference unsqueezes to [B,1,Tx] for its Conv1d)
        energy = batch['energy']

        # token-side row count (incl. the conv bank's extra row): trainer.TrainStep keeps weight gradients of
        # operands this short on the main stream (ops.GradSink.inline_rows)
        self.wgrad_inline_rows = x.shape[0] * (x.shape[1] + 1)
        self.wgrad_defer = True          # recurrences ahead: ops.GradSink.defer
        if self.training:
            self.step += 1
            self._bump_batchnorm_counters()

        # The three predictors are independent of the trunk (forward_tacotron.py:129-131 vs :133-159) and
        # their 128-step recurrences are latency-bound, so they run on a side HIP stream concurrently with the
        # trunk; autograd replays each backward node on the stream of its forward, so the overlap also holds
        # in backward.
        main = torch.cuda.current_stream()
        side = self._side_stream(x.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            dur_hat = self.dur_pred(x).squeeze(-1)
            pitch_hat = self.pitch_pred(x).transpose(1, 2)
            energy_hat = self.energy_pred(x).transpose(1, 2)
            hook = getattr(self, 'predictor_hook', None)   # trainer.TrainStep: the predictors' losses + backward, right here
            if hook is not None:
                hook({'dur': dur_hat, 'pitch': pitch_hat, 'energy': energy_hat})

        mel_cl, post_cl = self._trunk(x, dur, pitch, energy, mel_lens.to(device=x.device, dtype=torch.long))
        Tout = mel.size(2)
        x_post = ops.TransposePadFn.apply(post_cl, Tout, self.padding_value)
        x_mel = ops.TransposePadFn.apply(mel_cl, Tout, self.padding_value)
        main.wait_stream(side)
        H.rnn_note_join(main, side)
        for t in (dur_hat, pitch_hat, energy_hat):
            t.record_stream(main)
        return {'mel': x_mel, 'mel_post': x_post, 'dur': dur_hat, 'pitch': pitch_hat, 'energy': energy_hat}

    def _side_stream(self, device) -> 'torch.cuda.Stream':
        key = torch.device(device).index or 0
        if not hasattr(self, '_streams'):
            self._streams = {}
        if key not in self._streams:
            # high priority like the trainer's main stream: the predictors' kernels are small and many, behind the trunk's
            # 1000-workgroup GEMMs in a default-priority queue each of them waits for a free CU (0.25 ms of the step)
            self._streams[key] = torch.cuda.Stream(device=device, priority=_side_priority())
        return self._streams[key]

    def generate(self, x: torch.Tensor, alpha=1.0,
                 pitch_function: Callable[[torch.Tensor], torch.Tensor] = lambda x: x,
                 energy_function: Callable[[torch.Tensor], torch.Tensor] = lambda x: x) -> Dict[str, torch.Tensor]:
        self.eval()
        with torch.no_grad():
            return self._generate(x, alpha, pitch_function, energy_function)

    def generate_jit(self, x: torch.Tensor, alpha: float = 1.0, beta: float = 1.0) -> Dict[str, torch.Tensor]:
        """forward_tacotron.py:186-200: generate with the pitch scaled by beta.  Eager entry; the TorchScript surface
        (`torch.jit.script(model).generate_jit`, README.md:159-171 of the reference) is export.ScriptedForwardTacotron,
        which reaches this same path through the opaque operator torch.ops.fwdtaco.generate_jit."""
        with torch.no_grad():
            return self._generate(x, alpha, lambda p: ops.ScaleFn.apply(p, beta) if beta != 1.0 else p, lambda e: e)

    def __prepare_scriptable__(self):
        """torch.jit.script(model) compiles the flat-buffer twin (export.py), not this kernel-launching module tree."""
        from . import export
        return export.scriptable(self, self._ctor_kwargs)

    def _generate(self, x, alpha, pitch_function, energy_function):
        self._require_device(x)
        # The predictors only meet the trunk behind the prenet (forward_tacotron.py:168-189 runs them first, then the
        # prenet): they run on the side stream while the main stream does embedding + prenet CBHG (its 128-step GRU is
        # latency-bound: a single utterance spends 84 % of its time in recurrences), joined where their outputs are needed.
        import os
        main = torch.cuda.current_stream()
        side = self._side_stream(x.device) if os.environ.get('FT_GEN_OVERLAP', '1') == '1' else main
        side.wait_stream(main)
        with torch.cuda.stream(side):
            dur_hat = self.dur_pred(x, alpha=alpha).squeeze(2)
            pitch_hat = pitch_function(self.pitch_pred(x).transpose(1, 2))
            energy_hat = energy_function(self.energy_pred(x).transpose(1, 2))
        got = {}

        def late_inputs():
            main.wait_stream(side)
            H.rnn_note_join(main, side)
            for t in (dur_hat, pitch_hat, energy_hat):
                t.record_stream(main)
            if torch.sum(dur_hat.long()) <= 0:
                torch.fill_(dur_hat, value=2.)
            got['dur'] = dur_hat.contiguous()
            return (got['dur'], pitch_hat.reshape(x.shape[0], -1).contiguous(),
                    energy_hat.reshape(x.shape[0], -1).contiguous())

        mel_cl, post_cl = self._trunk(x, None, None, None, None, late_inputs=late_inputs)
        T = mel_cl.shape[1]
        return {'mel': H.transpose_pad_fwd(mel_cl, T, 0.0), 'mel_post': H.transpose_pad_fwd(post_cl, T, 0.0),
                'dur': got['dur'], 'pitch': pitch_hat, 'energy': energy_hat}

    def _generate_mel(self, x: torch.Tensor, dur_hat: torch.Tensor, pitch_hat: torch.Tensor,
                      energy_hat: torch.Tensor) -> Dict[str, torch.Tensor]:
        """forward_tacotron.py:205-234: mel generation from given durations [B,Tx], pitch and energy [B,1,Tx]; the LSTM
        runs over the padded length (no packing).  `dur_hat` is clamped in place like every LengthRegulator input."""
        self._require_device(x)
        dur_in = dur_hat.contiguous()
        mel_cl, post_cl = self._trunk(x, dur_in, pitch_hat.reshape(x.shape[0], -1).contiguous(),
                                      energy_hat.reshape(x.shape[0], -1).contiguous(), None)
        T = mel_cl.shape[1]
        return {'mel': H.transpose_pad_fwd(mel_cl, T, 0.0), 'mel_post': H.transpose_pad_fwd(post_cl, T, 0.0),
                'dur': dur_in, 'pitch': pitch_hat, 'energy': energy_hat}

    def get_step(self) -> int:
        return self.step.data.item()

    def _pad(self, x: torch.Tensor, max_len: int) -> torch.Tensor:
        """forward_tacotron.py:236-239 on a [B,C,T] tensor (kept for API parity; forward() fuses it)."""
        x = x[:, :, :max_len]
        return torch.nn.functional.pad(x, [0, max_len - x.size(2), 0, 0], 'constant', self.padding_value)

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> 'ForwardTacotron':
        model_config = config['forward_tacotron']['model']
        model_config['num_chars'] = config.get('num_chars', NUM_CHARS_DEFAULT)   # reference: len(phonemes)
        model_config['n_mels'] = config['dsp']['num_mels']
        return ForwardTacotron(**model_config)

    @classmethod
    def from_checkpoint(cls, path: Union[Path, str]) -> 'ForwardTacotron':
        checkpoint = torch.load(path, map_location=torch.device('cpu'), weights_only=True)
        model = ForwardTacotron.from_config(checkpoint['config'])
        model.load_state_dict(checkpoint['model'])
        return model


from . import export as _export  # noqa: E402,F401  (registers torch.ops.fwdtaco.generate_jit for torch.jit.load)
