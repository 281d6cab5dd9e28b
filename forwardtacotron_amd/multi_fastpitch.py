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
from .fastpitch import ForwardTransformer
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

    def __init__(self, num_chars: int,
                 durpred_dropout: float, durpred_d_model: int, durpred_n_heads: int, durpred_layers: int,
                 durpred_d_fft: int,
                 pitch_dropout: float, pitch_d_model: int, pitch_n_heads: int, pitch_layers: int, pitch_d_fft: int,
                 energy_dropout: float, energy_d_model: int, energy_n_heads: int, energy_layers: int,
                 energy_d_fft: int,
                 pitch_cond_d_model: int, pitch_cond_n_heads: int, pitch_cond_layers: int, pitch_cond_d_fft: int,
                 pitch_cond_dropout: float, pitch_cond_output_dims: int,
                 pitch_strength: float, energy_strength: float, d_model: int, conv1_kernel: int, conv2_kernel: int,
                 prenet_layers: int, prenet_heads: int, prenet_fft: int, prenet_dropout: float,
                 postnet_layers: int, postnet_heads: int, postnet_fft: int, postnet_dropout: float,
                 n_mels: int, speaker_emb_dims: int, padding_value=PAD_VALUE):
        super().__init__()
        self.padding_value = padding_value
        self.lr = LengthRegulator()
        common = dict(num_chars=num_chars, conv1_kernel=conv1_kernel, conv2_kernel=conv2_kernel,
                      speaker_emb_dims=speaker_emb_dims)
        self.dur_pred = ConditionalSeriesPredictor(d_model=durpred_d_model, n_heads=durpred_n_heads,
                                                   layers=durpred_layers, d_fft=durpred_d_fft,
                                                   dropout=durpred_dropout, **common)
        self.pitch_pred = ConditionalSeriesPredictor(d_model=pitch_d_model, n_heads=pitch_n_heads, layers=pitch_layers,
                                                     d_fft=pitch_d_fft, dropout=pitch_dropout, **common)
        self.pitch_cond_pred = SeriesPredictor(d_model=pitch_cond_d_model, n_heads=pitch_cond_n_heads,
                                               layers=pitch_cond_layers, d_fft=pitch_cond_d_fft,
                                               dropout=pitch_cond_dropout, out_dim=pitch_cond_output_dims, **common)
        self.energy_pred = SeriesPredictor(d_model=energy_d_model, n_heads=energy_n_heads, layers=energy_layers,
                                           d_fft=energy_d_fft, dropout=energy_dropout, **common)
        self.embedding = nn.Embedding(num_embeddings=num_chars, embedding_dim=d_model)
        wide = d_model + speaker_emb_dims
        self.prenet = ForwardTransformer(heads=prenet_heads, dropout=prenet_dropout, conv1_kernel=conv1_kernel,
                                         conv2_kernel=conv2_kernel, d_model=wide, d_fft=prenet_fft,
                                         layers=prenet_layers)
        self.postnet = ForwardTransformer(heads=postnet_heads, dropout=postnet_dropout, conv1_kernel=conv1_kernel,
                                          conv2_kernel=conv2_kernel, d_model=wide, d_fft=postnet_fft,
                                          layers=postnet_layers)
        self.lin = nn.Linear(wide, n_mels)
        self.register_buffer('step', torch.zeros(1, dtype=torch.long))
        self.pitch_strength = pitch_strength
        self.energy_strength = energy_strength
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
        dur_hat = self.dur_pred(x, pitch_cond, semb, src_pad_mask=len_mask).squeeze(-1)
        pitch_hat = self.pitch_pred(x, pitch_cond, semb, src_pad_mask=len_mask).transpose(1, 2)
        pitch_cond_hat = self.pitch_cond_pred(x, semb, src_pad_mask=len_mask)
        energy_hat = self.energy_pred(x, semb, src_pad_mask=len_mask).transpose(1, 2)
        mel_cl = self._mel(x, semb, len_mask, dur, batch['pitch'], batch['energy'],
                           mel_lens.to(device=x.device, dtype=torch.long))
        x_mel = ops.TransposePadFn.apply(mel_cl, mel.size(2), self.padding_value)
        return {'mel': x_mel, 'mel_post': x_mel, 'pitch_cond': pitch_cond_hat, 'dur': dur_hat, 'pitch': pitch_hat,
                'energy': energy_hat}

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
