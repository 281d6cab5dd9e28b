"""Synthetic LJSpeech-shaped batches with the ForwardCollator layout (utils/dataset.py:239-263,
tests/test_collator.py:40-65 of the reference): x zero-padded int64, mel padded to max(mel_len)+1 with
-11.5129, dur/pitch/energy zero-padded fp32, x_len/mel_len int64.  Seeded; no dataset needed."""
from typing import Dict

import torch

PAD_VALUE = -11.5129

SINGLESPEAKER_MODEL = dict(   # configs/singlespeaker.yaml:98-128 of the reference (+ num_chars / n_mels)
    embed_dims=256, series_embed_dims=64, num_chars=135,
    durpred_conv_dims=256, durpred_rnn_dims=64, durpred_dropout=0.5,
    pitch_conv_dims=256, pitch_rnn_dims=128, pitch_dropout=0.5, pitch_strength=1.0,
    energy_conv_dims=256, energy_rnn_dims=64, energy_dropout=0.5, energy_strength=1.0,
    rnn_dims=512, prenet_dims=256, prenet_k=16, postnet_num_highways=4,
    prenet_dropout=0.5, postnet_dims=256, postnet_k=8, prenet_num_highways=4,
    postnet_dropout=0.0, n_mels=80)
SINGLESPEAKER_TRAIN = dict(dur_loss_factor=0.1, pitch_loss_factor=0.1, energy_loss_factor=0.1,
                           pitch_zoneout=0.0, energy_zoneout=0.0, clip_grad_norm=1.0)


FASTPITCH_MODEL = dict(       # configs/singlespeaker.yaml:152-187 of the reference (+ num_chars / n_mels)
    num_chars=135,
    durpred_d_model=128, durpred_n_heads=2, durpred_layers=4, durpred_d_fft=128, durpred_dropout=0.5,
    pitch_d_model=128, pitch_n_heads=2, pitch_layers=4, pitch_d_fft=128, pitch_dropout=0.5, pitch_strength=1.0,
    energy_d_model=128, energy_n_heads=2, energy_layers=4, energy_d_fft=128, energy_dropout=0.5, energy_strength=1.0,
    d_model=256, conv1_kernel=9, conv2_kernel=1,
    prenet_layers=4, prenet_heads=2, prenet_fft=1024, prenet_dropout=0.1,
    postnet_layers=4, postnet_heads=2, postnet_fft=1024, postnet_dropout=0.1, n_mels=80)

MULTISPEAKER_MODEL = dict(    # configs/multispeaker.yaml:100-138 of the reference (+ num_chars / n_mels)
    SINGLESPEAKER_MODEL, series_embed_dims=128, durpred_rnn_dims=128, pitch_rnn_dims=256,
    pitch_cond_conv_dims=256, pitch_cond_rnn_dims=128, pitch_cond_dropout=0.5,
    speaker_emb_dims=256, pitch_cond_emb_dims=4, pitch_cond_categorical_dims=3)


def fastpitch_train_flops(n_tok: int, n_frm: int, Tx: int, Tm: int) -> float:
    """valid tokens/frames, attention terms on the padded lengths, train = 3 x forward."""
    m_tok = 3 * (917_504 + 1_024 * Tx + 128) + (11_534_336 + 2_048 * Tx) + 1_536
    m_frm = 11_534_336 + 2_048 * Tm + 20_480
    return 3.0 * 2.0 * (m_tok * n_tok + m_frm * n_frm)


def synthetic_batch(B: int = 32, Tmax: int = 128, n_mels: int = 80, num_chars: int = 135, max_dur: int = 12,
                    seed: int = 0) -> Dict[str, torch.Tensor]:
    """SURVEY.md section 8d recipe: B=32/Tmax=128/seed 0 -> 3,218 tokens, 19,320 frames, Tm=841."""
    g = torch.Generator().manual_seed(seed)
    x_len = torch.randint(Tmax // 2, Tmax + 1, (B,), generator=g)
    x_len[0] = Tmax
    x = torch.zeros(B, Tmax, dtype=torch.long)
    dur = torch.zeros(B, Tmax)
    for b in range(B):
        L = int(x_len[b])
        x[b, :L] = torch.randint(1, num_chars, (L,), generator=g)
        dur[b, :L] = torch.randint(1, max_dur, (L,), generator=g).float()
    mel_len = dur.sum(1).long()
    Tm = int(mel_len.max())
    mel = torch.full((B, n_mels, Tm + 1), PAD_VALUE)
    for b in range(B):
        n = int(mel_len[b])
        mel[b, :, :n] = torch.randn(n_mels, n, generator=g) * 2 - 5
    pitch = torch.randn(B, Tmax, generator=g) * (x > 0)
    energy = torch.rand(B, Tmax, generator=g) * (x > 0)
    return {'x': x, 'mel': mel, 'dur': dur, 'x_len': x_len, 'mel_len': mel_len, 'pitch': pitch, 'energy': energy}


def to_device(batch: Dict[str, torch.Tensor], device) -> Dict[str, torch.Tensor]:
    """trainer/common.py:95-101"""
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}


def train_flops(n_tok: int, n_frm: int) -> float:
    """Algorithmic FLOPs of one train step (valid tokens/frames only, 2 FLOP/MAC, train = 3 x forward),
    singlespeaker ForwardTacotron: SURVEY.md section 8d / BASELINE.md section 3."""
    return 3.0 * 2.0 * (16_387_584 * n_tok + 8_019_968 * n_frm)
