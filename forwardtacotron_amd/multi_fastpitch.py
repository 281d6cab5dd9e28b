"""MI355X-native MultiFastPitch (speaker-conditioned FastPitch): drop-in for models/multi_fast_pitch.py:14-328 -- same
constructor kwargs, batch-dict forward()/generate() and state_dict.  Composition of the FastPitch transformer blocks
(forwardtacotron_amd.fastpitch) with the speaker / pitch_cond column concat of the multispeaker ForwardTacotron.
"""
from pathlib import Path
from typing import Any, Callable, Dict, Optional, Union

import torch
import torch.nn as nn

from . import _lib
from . import hip as H
from . import ops
from .fastpitch import ForwardTransformer, precision_scoped
from .model import LengthRegulator, NUM_CHARS_DEFAULT, PAD_VALUE


class SeriesPredictor(nn.Module):
    """multi_fast_pitch.py:14-49: embedding ++ speaker embedding -> ForwardTransformer -> Linear(out_dim)."""

    def __init__(self, num_chars: int, d_model: int, n_heads: int, d_fft: int, layers: int, conv1_kernel: int,
                 conv2_kernel: int, speaker_emb_dims: int, dropout=0.1, out_dim: int = 1):
        super().__init__()
        self.embedding = nn.Embedding(num_chars, d_model)
        self.transformer = ForwardTransformer(heads=n_heads, dropout=dropout, d_model=d_model + speaker_emb_dims,
                                              d_fft=d_fft, conv1_kernel=conv1_kernel, conv2_kernel=conv2_kernel,
                                              layers=layers)
        self.lin = nn.Linear(d_model + speaker_emb_dims, out_dim)

    def forward(self, x: torch.Tensor, speaker_emb: torch.Tensor, src_pad_mask: Optional[torch.Tensor] = None,
                alpha: float = 1.0) -> torch.Tensor:
        B, T = x.shape
        x = ops.EmbeddingFn.apply(x, self.embedding.weight)
        x = ops.ConcatColsFn.apply(x, None, speaker_emb, B, T, False)
        x = self.transformer(x, src_pad_mask=src_pad_mask)
        x = ops.LinearFn.apply(x, self.lin.weight, self.lin.bias)
        if alpha != 1.0:
            x = ops.ScaleFn.apply(x, 1.0 / alpha)
        return x


class ConditionalSeriesPredictor(nn.Module):
    """multi_fast_pitch.py:52-90: embedding ++ conditional embedding ++ speaker embedding -> ..."""

    def __init__(self, num_chars: int, d_model: int, n_heads: int, d_fft: int, layers: int, conv1_kernel: int,
                 conv2_kernel: int, speaker_emb_dims: int, cond_emb_size: int = 4, cond_emb_dims: int = 8, dropout=0.1):
        super().__init__()
        self.embedding = nn.Embedding(num_chars, d_model)
        self.conditional_embedding = nn.Embedding(cond_emb_size, cond_emb_dims)
        self.transformer = ForwardTransformer(heads=n_heads, dropout=dropout,
                                              d_model=d_model + speaker_emb_dims + cond_emb_dims, d_fft=d_fft,
                                              conv1_kernel=conv1_kernel, conv2_kernel=conv2_kernel, layers=layers)
        self.lin = nn.Linear(d_model + speaker_emb_dims + cond_emb_dims, 1)

    def forward(self, x: torch.Tensor, x_cond: torch.Tensor, speaker_emb: torch.Tensor,
                src_pad_mask: Optional[torch.Tensor] = None, alpha: float = 1.0) -> torch.Tensor:
        B, T = x.shape
        e = ops.EmbeddingFn.apply(x, self.embedding.weight)
        c = ops.EmbeddingFn.apply(x_cond, self.conditional_embedding.weight)
        x = ops.ConcatColsFn.apply(e, c, speaker_emb, B, T, False)
        x = self.transformer(x, src_pad_mask=src_pad_mask)
        x = ops.LinearFn.apply(x, self.lin.weight, self.lin.bias)
        if alpha != 1.0:
            x = ops.ScaleFn.apply(x, 1.0 / alpha)
        return x


class MultiFastPitch(nn.Module):
    """Drop-in for models/multi_fast_pitch.py:93-328."""

    # Constructor keywords = the keys of config['multi_fast_pitch']['model'] (+ num_chars, n_mels), exactly the
    # reference's (multi_fast_pitch.py:95-133).  Four predictors share one hyper-parameter pattern
    # `<prefix>_{d_model,n_heads,layers,d_fft,dropout}`, the two transformers `<prefix>_{layers,heads,fft,dropout}`.
    _PREDICTORS = ('durpred', 'pitch', 'energy', 'pitch_cond')
    _TRUNKS = ('prenet', 'postnet')
    _SCALARS = ('num_chars', 'n_mels', 'speaker_emb_dims', 'd_model', 'conv1_kernel', 'conv2_kernel',
                'pitch_strength', 'energy_strength', 'pitch_cond_output_dims')

    @classmethod
    def _expected(cls):
        keys = list(cls._SCALARS)
        keys += [f'{p}_{k}' for p in cls._PREDICTORS for k in ('d_model', 'n_heads', 'layers', 'd_fft', 'dropout')]
        keys += [f'{p}_{k}' for p in cls._TRUNKS for k in ('layers', 'heads', 'fft', 'dropout')]
        return keys

    def __init__(self, padding_value=PAD_VALUE, **hp):
        super().__init__()
        want = self._expected()
        missing = [k for k in want if k not in hp]
        extra = [k for k in hp if k not in want]
        if missing or extra:
            raise TypeError(f'MultiFastPitch(): missing {missing}, unexpected {extra}')
        self.padding_value = padding_value
        self.matmul_dtype = 'fp32'          # or 'bf16': see fastpitch.FastPitch
        self.lr = LengthRegulator()
        # predictor branches share no graph node with the trunk in training (trainer.TrainStep may run their backward as a
        # stage of its own)
        self.independent_predictors = True
        shared = {k: hp[k] for k in ('num_chars', 'conv1_kernel', 'conv2_kernel', 'speaker_emb_dims')}

        def predictor(kind, prefix, **more):
            return kind(d_model=hp[prefix + '_d_model'], n_heads=hp[prefix + '_n_heads'], layers=hp[prefix + '_layers'],
                        d_fft=hp[prefix + '_d_fft'], dropout=hp[prefix + '_dropout'], **shared, **more)

        wide = hp['d_model'] + hp['speaker_emb_dims']

        def trunk(prefix):
            return ForwardTransformer(d_model=wide, heads=hp[prefix + '_heads'], d_fft=hp[prefix + '_fft'],
                                      layers=hp[prefix + '_layers'], dropout=hp[prefix + '_dropout'],
                                      conv1_kernel=hp['conv1_kernel'], conv2_kernel=hp['conv2_kernel'])

        # registration order = the reference's (it fixes the state_dict key order)
        self.dur_pred = predictor(ConditionalSeriesPredictor, 'durpred')
        self.pitch_pred = predictor(ConditionalSeriesPredictor, 'pitch')
        self.pitch_cond_pred = predictor(SeriesPredictor, 'pitch_cond', out_dim=hp['pitch_cond_output_dims'])
        self.energy_pred = predictor(SeriesPredictor, 'energy')
        self.embedding = nn.Embedding(hp['num_chars'], hp['d_model'])
        self.prenet = trunk('prenet')
        self.postnet = trunk('postnet')
        self.lin = nn.Linear(wide, hp['n_mels'])
        self.register_buffer('step', torch.zeros(1, dtype=torch.long))
        self.pitch_strength = hp['pitch_strength']
        self.energy_strength = hp['energy_strength']
        self.pitch_proj = nn.Conv1d(1, wide, kernel_size=3, padding=1)
        self.energy_proj = nn.Conv1d(1, wide, kernel_size=3, padding=1)

    def __repr__(self):
        return f'MultiFastPitch, num params: {sum(p.numel() for p in self.parameters())}'

    def _require_device(self, t: torch.Tensor) -> None:
        if not t.is_cuda or not self.embedding.weight.is_cuda:
            raise _lib.FtError('MultiFastPitch runs on an MI355X (HIP) device only; there is no CPU fallback')

    def _mel(self, x_idx, semb, tok_mask, dur, pitch, energy, frame_lens: Optional[torch.Tensor]):
        B, Tx = x_idx.shape
        x = ops.EmbeddingFn.apply(x_idx, self.embedding.weight)
        x = ops.ConcatColsFn.apply(x, None, semb, B, Tx, False)
        x = self.prenet(x, src_pad_mask=tok_mask)
        x = ops.CondAddFn.apply(x, pitch, energy, self.pitch_proj.weight, self.pitch_proj.bias,
                                self.energy_proj.weight, self.energy_proj.bias, self.pitch_strength,
                                self.energy_strength, False)
        x = self.lr(x, dur)
        frame_mask = None
        if frame_lens is not None:          # multi_fast_pitch.py:230-232
            frame_mask = torch.arange(x.shape[1], device=x.device).unsqueeze(0) >= frame_lens.unsqueeze(1)
        x = self.postnet(x, src_pad_mask=frame_mask)
        return ops.LinearFn.apply(x, self.lin.weight, self.lin.bias)

    @precision_scoped
    def forward(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        x = batch['x']
        mel = batch['mel']
        dur = batch['dur']
        semb = batch['speaker_emb'].contiguous()
        mel_lens = batch['mel_len']
        pitch_cond = batch['pitch_cond']
        self._require_device(x)
        if self.training:
            self.step += 1
        len_mask = x == 0
        # predictors on a side HIP stream, concurrently with the frame-side trunk (see FastPitch.forward)
        main = torch.cuda.current_stream()
        key = x.device.index or 0
        if not hasattr(self, '_streams'):
            self._streams = {}
        if key not in self._streams:
            from .model import _side_priority
            self._streams[key] = torch.cuda.Stream(device=x.device, priority=_side_priority())
        side = self._streams[key]
        side.wait_stream(main)
        with torch.cuda.stream(side):
            dur_hat = self.dur_pred(x, pitch_cond, semb, src_pad_mask=len_mask).squeeze(-1)
            pitch_hat = self.pitch_pred(x, pitch_cond, semb, src_pad_mask=len_mask).transpose(1, 2)
            pitch_cond_hat = self.pitch_cond_pred(x, semb, src_pad_mask=len_mask)
            energy_hat = self.energy_pred(x, semb, src_pad_mask=len_mask).transpose(1, 2)
            hook = getattr(self, 'predictor_hook', None)   # trainer.TrainStep: the predictors' losses + backward, right here
            if hook is not None:
                hook({'dur': dur_hat, 'pitch': pitch_hat, 'energy': energy_hat, 'pitch_cond': pitch_cond_hat})
        mel_cl = self._mel(x, semb, len_mask, dur, batch['pitch'], batch['energy'],
                           mel_lens.to(device=x.device, dtype=torch.long))
        x_mel = ops.TransposePadFn.apply(mel_cl, mel.size(2), self.padding_value)
        main.wait_stream(side)
        for t in (dur_hat, pitch_hat, pitch_cond_hat, energy_hat):
            t.record_stream(main)
        return {'mel': x_mel, 'mel_post': x_mel, 'pitch_cond': pitch_cond_hat, 'dur': dur_hat, 'pitch': pitch_hat,
                'energy': energy_hat}

    @precision_scoped
    def generate(self, x: torch.Tensor, speaker_emb: torch.Tensor, alpha=1.0,
                 pitch_function: Callable[[torch.Tensor], torch.Tensor] = lambda x: x,
                 energy_function: Callable[[torch.Tensor], torch.Tensor] = lambda x: x) -> Dict[str, torch.Tensor]:
        self.eval()
        with torch.no_grad():
            self._require_device(x)
            speaker_emb = speaker_emb.contiguous()
            B = x.shape[0]
            # reference quirks kept (multi_fast_pitch.py:255-263): predictors run unmasked, the pitch_cond logits are
            # divided by alpha, and the argmax/unsqueeze chain only works for B = 1
            pitch_cond_hat = self.pitch_cond_pred(x, speaker_emb, alpha=alpha).squeeze(-1)
            pitch_cond_hat = torch.argmax(pitch_cond_hat.squeeze(), dim=1).long().unsqueeze(0)
            dur_hat = self.dur_pred(x, pitch_cond_hat, speaker_emb, alpha=alpha).squeeze(2)
            if torch.sum(dur_hat.long()) <= 0:
                torch.fill_(dur_hat, value=2.)
            pitch_hat = pitch_function(self.pitch_pred(x, pitch_cond_hat, speaker_emb).transpose(1, 2))
            energy_hat = energy_function(self.energy_pred(x, speaker_emb).transpose(1, 2))
            dur_in = dur_hat.contiguous()
            mel_cl = self._mel(x, speaker_emb, x == 0, dur_in, pitch_hat.reshape(B, -1).contiguous(),
                               energy_hat.reshape(B, -1).contiguous(), None)
            m = H.transpose_pad_fwd(mel_cl, mel_cl.shape[1], 0.0)
            return {'mel': m, 'mel_post': m, 'dur': dur_in, 'pitch_cond': pitch_cond_hat, 'pitch': pitch_hat,
                    'energy': energy_hat}

    def pad(self, x: torch.Tensor, max_len: int) -> torch.Tensor:
        x = x[:, :, :max_len]
        return torch.nn.functional.pad(x, [0, max_len - x.size(2), 0, 0], 'constant', self.padding_value)

    def get_step(self) -> int:
        return self.step.data.item()

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> 'MultiFastPitch':
        model_config = config['multi_fast_pitch']['model']
        model_config['num_chars'] = config.get('num_chars', NUM_CHARS_DEFAULT)
        model_config['n_mels'] = config['dsp']['num_mels']
        return MultiFastPitch(**model_config)

    @classmethod
    def from_checkpoint(cls, path: Union[Path, str]) -> 'MultiFastPitch':
        checkpoint = torch.load(path, map_location=torch.device('cpu'), weights_only=True)
        model = MultiFastPitch.from_config(checkpoint['config'])
        model.load_state_dict(checkpoint['model'])
        return model
