"""MI355X-native MultiForwardTacotron (speaker-embedding conditioned variant): drop-in for
models/multi_forward_tacotron.py:14-323 -- same constructor kwargs, batch-dict forward()/generate(), and
353-entry state_dict.  Built from the same HIP ops as forwardtacotron_amd.model plus the speaker concat,
the conditional predictors and the 3-class pitch_cond head.
"""
from pathlib import Path
from typing import Any, Callable, Dict, Optional, Union

import torch
import torch.nn as nn

from . import hip as H
from . import ops
from .model import (BatchNormConv, CBHG, GRU, LSTM, LengthRegulator, NUM_CHARS_DEFAULT, PAD_VALUE, _dropout,
                    _side_priority, regulate_and_decode)


class SeriesPredictor(nn.Module):
    """multi_forward_tacotron.py:14-50: embedding ++ speaker embedding -> 3 BatchNormConv -> biGRU -> Linear."""

    def __init__(self, num_chars: int, emb_dim: int = 64, conv_dims: int = 256, rnn_dims: int = 64,
                 dropout: float = 0.5, speaker_emb_dims: int = 256, out_dim: int = 1):
        super().__init__()
        self.embedding = nn.Embedding(num_chars, emb_dim)
        self.convs = nn.ModuleList([
            BatchNormConv(emb_dim + speaker_emb_dims, conv_dims, 5, relu=True),
            BatchNormConv(conv_dims, conv_dims, 5, relu=True),
            BatchNormConv(conv_dims, conv_dims, 5, relu=True),
        ])
        self.rnn = GRU(conv_dims, rnn_dims)
        self.lin = nn.Linear(2 * rnn_dims, out_dim)
        self.dropout = dropout

    def forward(self, x: torch.Tensor, semb: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
        B, T = x.shape
        x = ops.EmbeddingFn.apply(x, self.embedding.weight)
        x = ops.ConcatColsFn.apply(x, None, semb, B, T, False)
        for conv in self.convs:
            x = conv(x)
            x = _dropout(x, self.dropout, self.training)
        x = self.rnn(x, time_major_out=True)
        x = ops.LinearFn.apply(x, self.lin.weight, self.lin.bias, B)
        if alpha != 1.0:
            x = ops.ScaleFn.apply(x, 1.0 / alpha)
        return x


class ConditionalSeriesPredictor(nn.Module):
    """multi_forward_tacotron.py:53-93: embedding ++ pitch_cond embedding ++ speaker embedding -> ..."""

    def __init__(self, num_chars: int, emb_dim: int = 64, cond_emb_size: int = 4, cond_emb_dims: int = 8,
                 conv_dims: int = 256, rnn_dims: int = 64, dropout: float = 0.5, speaker_emb_dims: int = 256):
        super().__init__()
        self.embedding = nn.Embedding(num_chars, emb_dim)
        self.pitch_cond_embedding = nn.Embedding(cond_emb_size, cond_emb_dims)
        self.convs = nn.ModuleList([
            BatchNormConv(emb_dim + cond_emb_dims + speaker_emb_dims, conv_dims, 5, relu=True),
            BatchNormConv(conv_dims, conv_dims, 5, relu=True),
            BatchNormConv(conv_dims, conv_dims, 5, relu=True),
        ])
        self.rnn = GRU(conv_dims, rnn_dims)
        self.lin = nn.Linear(2 * rnn_dims, 1)
        self.dropout = dropout

    def forward(self, x: torch.Tensor, x_cond: torch.Tensor, speaker_emb: torch.Tensor,
                alpha: float = 1.0) -> torch.Tensor:
        B, T = x.shape
        e = ops.EmbeddingFn.apply(x, self.embedding.weight)
        c = ops.EmbeddingFn.apply(x_cond, self.pitch_cond_embedding.weight)
        x = ops.ConcatColsFn.apply(e, c, speaker_emb, B, T, False)
        for conv in self.convs:
            x = conv(x)
            x = _dropout(x, self.dropout, self.training)
        x = self.rnn(x, time_major_out=True)
        x = ops.LinearFn.apply(x, self.lin.weight, self.lin.bias, B)
        if alpha != 1.0:
            x = ops.ScaleFn.apply(x, 1.0 / alpha)
        return x


class MultiForwardTacotron(nn.Module):
    """Drop-in for models/multi_forward_tacotron.py:96-323."""

    # Constructor keywords = the keys of config['multi_forward_tacotron']['model'] (+ num_chars, n_mels), exactly the
    # reference's (multi_forward_tacotron.py:98-129).
    _KEYS = ('embed_dims', 'series_embed_dims', 'num_chars', 'rnn_dims', 'n_mels', 'speaker_emb_dims',
             'pitch_cond_emb_dims', 'pitch_cond_categorical_dims', 'pitch_strength', 'energy_strength',
             'prenet_dims', 'prenet_k', 'prenet_num_highways', 'prenet_dropout',
             'postnet_dims', 'postnet_k', 'postnet_num_highways', 'postnet_dropout') + tuple(
        f'{p}_{k}' for p in ('durpred', 'pitch', 'pitch_cond', 'energy') for k in ('conv_dims', 'rnn_dims', 'dropout'))

    def __init__(self, padding_value=PAD_VALUE, **hp):
        super().__init__()
        missing = [k for k in self._KEYS if k not in hp]
        extra = [k for k in hp if k not in self._KEYS]
        if missing or extra:
            raise TypeError(f'MultiForwardTacotron(): missing {missing}, unexpected {extra}')
        self.rnn_dims = hp['rnn_dims']
        self.padding_value = padding_value
        E, P, Q, S = hp['embed_dims'], hp['prenet_dims'], hp['postnet_dims'], hp['speaker_emb_dims']
        self.embedding = nn.Embedding(hp['num_chars'], E)
        self.lr = LengthRegulator()
        # predictor branches share no graph node with the trunk in training (trainer.TrainStep may run their backward as a
        # stage of its own)
        self.independent_predictors = True

        def predictor(kind, prefix, **more):
            # NB (reference quirk, multi_forward_tacotron.py:135-157): speaker_emb_dims is NOT forwarded to the
            # predictors, they keep their default of 256.
            return kind(num_chars=hp['num_chars'], emb_dim=hp['series_embed_dims'], conv_dims=hp[prefix + '_conv_dims'],
                        rnn_dims=hp[prefix + '_rnn_dims'], dropout=hp[prefix + '_dropout'], **more)

        # registration order = the reference's (it fixes the state_dict key order)
        self.dur_pred = predictor(ConditionalSeriesPredictor, 'durpred', cond_emb_dims=hp['pitch_cond_emb_dims'])
        self.pitch_cond_pred = predictor(SeriesPredictor, 'pitch_cond', out_dim=hp['pitch_cond_categorical_dims'])
        self.pitch_pred = predictor(ConditionalSeriesPredictor, 'pitch', cond_emb_dims=hp['pitch_cond_emb_dims'])
        self.energy_pred = predictor(SeriesPredictor, 'energy')
        self.prenet = CBHG(K=hp['prenet_k'], in_channels=E, channels=P, proj_channels=[P, E],
                           num_highways=hp['prenet_num_highways'], dropout=hp['prenet_dropout'])
        self.lstm = LSTM(2 * P + S, hp['rnn_dims'])
        self.lin = nn.Linear(2 * hp['rnn_dims'], hp['n_mels'])
        self.register_buffer('step', torch.zeros(1, dtype=torch.long))
        self.postnet = CBHG(K=hp['postnet_k'], in_channels=hp['n_mels'], channels=Q, proj_channels=[Q, hp['n_mels']],
                            num_highways=hp['postnet_num_highways'], dropout=hp['postnet_dropout'])
        self.post_proj = nn.Linear(2 * Q, hp['n_mels'], bias=False)
        self.pitch_strength = hp['pitch_strength']
        self.energy_strength = hp['energy_strength']
        self.pitch_proj = nn.Conv1d(1, 2 * P + S, kernel_size=3, padding=1)
        self.energy_proj = nn.Conv1d(1, 2 * P + S, kernel_size=3, padding=1)
        self._nbt_flat = None

    def __repr__(self):
        return f'MultiForwardTacotron, num params: {sum(p.numel() for p in self.parameters())}'

    def _require_device(self, t: torch.Tensor) -> None:
        if not t.is_cuda or not self.embedding.weight.is_cuda:
            raise H._lib.FtError('MultiForwardTacotron runs on an MI355X (HIP) device only; there is no CPU fallback')

    def _bump_batchnorm_counters(self) -> None:
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

    def _trunk(self, x, semb, dur, pitch, energy, mel_lens: Optional[torch.Tensor], late_inputs=None):
        """late_inputs (inference): delivers (dur, pitch, energy) once the prenet has been enqueued (see
        model.ForwardTacotron._trunk: the predictors run beside the prenet on the side stream)"""
        B, Tx = x.shape
        x = ops.EmbeddingFn.apply(x, self.embedding.weight)
        x = self.prenet(x, time_major_out=True)                                  # [Tx,B,2P]
        if late_inputs is not None:
            dur, pitch, energy = late_inputs()
        x = ops.ConcatColsFn.apply(x, None, semb, B, Tx, True)                   # [B,Tx,2P+S]
        x = ops.CondAddFn.apply(x, pitch, energy, self.pitch_proj.weight, self.pitch_proj.bias,
                                self.energy_proj.weight, self.energy_proj.bias, self.pitch_strength,
                                self.energy_strength, False)
        x = regulate_and_decode(self, x, dur, mel_lens)
        mel = ops.LinearFn.apply(x, self.lin.weight, self.lin.bias)
        post = self.postnet(mel, time_major_out=True)
        post = ops.LinearFn.apply(post, self.post_proj.weight, None, B)
        return mel, post

    def forward(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        x = batch['x']
        mel = batch['mel']
        dur = batch['dur']
        semb = batch['speaker_emb'].contiguous()
        mel_lens = batch['mel_len']
        pitch = batch['pitch']
        pitch_cond = batch['pitch_cond']
        energy = batch['energy']
        self._require_device(x)
        # token-side row count (incl. the conv bank's extra row): trainer.TrainStep keeps weight gradients of
        # operands this short on the main stream (ops.GradSink.inline_rows)
        self.wgrad_inline_rows = x.shape[0] * (x.shape[1] + 1)
        self.wgrad_defer = True          # recurrences ahead: ops.GradSink.defer
        if self.training:
            self.step += 1
            self._bump_batchnorm_counters()

        # the four predictors are independent of the trunk in training (it consumes the batch's targets,
        # multi_forward_tacotron.py:183-213) and their 128-step recurrences are latency-bound: side HIP stream,
        # concurrently with the trunk, as in the single-speaker model (autograd replays each backward node on the
        # stream of its forward)
        main = torch.cuda.current_stream()
        side = self._side_stream(x.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            pitch_cond_hat = self.pitch_cond_pred(x, semb).squeeze(-1)           # [B,Tx,3]
            dur_hat = self.dur_pred(x, pitch_cond, semb).squeeze(-1)
            pitch_hat = self.pitch_pred(x, pitch_cond, semb).transpose(1, 2)
            energy_hat = self.energy_pred(x, semb).transpose(1, 2)
            hook = getattr(self, 'predictor_hook', None)   # trainer.TrainStep: the predictors' losses + backward, right here
            if hook is not None:
                hook({'dur': dur_hat, 'pitch': pitch_hat, 'energy': energy_hat, 'pitch_cond': pitch_cond_hat})

        mel_cl, post_cl = self._trunk(x, semb, dur, pitch, energy, mel_lens.to(device=x.device, dtype=torch.long))
        Tout = mel.size(2)
        x_post = ops.TransposePadFn.apply(post_cl, Tout, self.padding_value)
        x_mel = ops.TransposePadFn.apply(mel_cl, Tout, self.padding_value)
        main.wait_stream(side)
        H.rnn_note_join(main, side)
        for t in (pitch_cond_hat, dur_hat, pitch_hat, energy_hat):
            t.record_stream(main)
        return {'mel': x_mel, 'mel_post': x_post, 'dur': dur_hat, 'pitch': pitch_hat, 'energy': energy_hat,
                'pitch_cond': pitch_cond_hat}

    def generate(self, x: torch.Tensor, speaker_emb: torch.Tensor, alpha=1.0,
                 pitch_function: Callable[[torch.Tensor], torch.Tensor] = lambda x: x,
                 energy_function: Callable[[torch.Tensor], torch.Tensor] = lambda x: x) -> Dict[str, torch.Tensor]:
        self.eval()
        with torch.no_grad():
            self._require_device(x)
            speaker_emb = speaker_emb.contiguous()
            # the four predictors beside embedding + prenet CBHG (side stream), joined where the trunk needs them
            import os
            main = torch.cuda.current_stream()
            side = self._side_stream(x.device) if os.environ.get('FT_GEN_OVERLAP', '1') == '1' else main
            side.wait_stream(main)
            with torch.cuda.stream(side):
                pitch_cond_hat = self.pitch_cond_pred(x, speaker_emb).squeeze(-1)
                pitch_cond_hat = torch.argmax(pitch_cond_hat.squeeze(), dim=1).long().unsqueeze(0)
                dur_hat = self.dur_pred(x, pitch_cond_hat, speaker_emb, alpha=alpha).squeeze(-1)
                pitch_hat = pitch_function(self.pitch_pred(x, pitch_cond_hat, speaker_emb).transpose(1, 2))
                energy_hat = energy_function(self.energy_pred(x, speaker_emb).transpose(1, 2))
            B = x.shape[0]
            got = {}

            def late_inputs():
                main.wait_stream(side)
                H.rnn_note_join(main, side)
                for t in (pitch_cond_hat, dur_hat, pitch_hat, energy_hat):
                    t.record_stream(main)
                if torch.sum(dur_hat.long()) <= 0:
                    torch.fill_(dur_hat, value=2.)
                got['dur'] = dur_hat.contiguous()
                return got['dur'], pitch_hat.reshape(B, -1).contiguous(), energy_hat.reshape(B, -1).contiguous()

            mel_cl, post_cl = self._trunk(x, speaker_emb, None, None, None, None, late_inputs=late_inputs)
            T = mel_cl.shape[1]
            return {'mel': H.transpose_pad_fwd(mel_cl, T, 0.0), 'mel_post': H.transpose_pad_fwd(post_cl, T, 0.0),
                    'dur': got['dur'], 'pitch': pitch_hat, 'energy': energy_hat,
                    'pitch_cond': pitch_cond_hat.unsqueeze(1)}

    def _side_stream(self, device) -> 'torch.cuda.Stream':
        key = torch.device(device).index or 0
        if not hasattr(self, '_streams'):
            self._streams = {}
        if key not in self._streams:
            self._streams[key] = torch.cuda.Stream(device=device, priority=_side_priority())
        return self._streams[key]

    def get_step(self) -> int:
        return self.step.data.item()

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> 'MultiForwardTacotron':
        model_config = config['multi_forward_tacotron']['model']
        model_config['num_chars'] = config.get('num_chars', NUM_CHARS_DEFAULT)
        model_config['n_mels'] = config['dsp']['num_mels']
        return MultiForwardTacotron(**model_config)

    @classmethod
    def from_checkpoint(cls, path: Union[Path, str]) -> 'MultiForwardTacotron':
        checkpoint = torch.load(path, map_location=torch.device('cpu'), weights_only=True)
        model = MultiForwardTacotron.from_config(checkpoint['config'])
        model.load_state_dict(checkpoint['model'])
        return model
