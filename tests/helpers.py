"""Shared test helpers (CPU side)."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

TINY = dict(embed_dims=16, series_embed_dims=8, num_chars=135,
            durpred_conv_dims=16, durpred_rnn_dims=8, durpred_dropout=0.0,
            pitch_conv_dims=16, pitch_rnn_dims=12, pitch_dropout=0.0, pitch_strength=1.0,
            energy_conv_dims=16, energy_rnn_dims=8, energy_dropout=0.0, energy_strength=0.5,
            rnn_dims=20, prenet_dims=16, prenet_k=4, postnet_num_highways=2,
            prenet_dropout=0.0, postnet_dims=12, postnet_k=3, prenet_num_highways=2,
            postnet_dropout=0.0, n_mels=10)

# deliberately awkward sizes (nothing a multiple of 4) for edge coverage
ODD = dict(embed_dims=10, series_embed_dims=6, num_chars=135,
           durpred_conv_dims=7, durpred_rnn_dims=5, durpred_dropout=0.0,
           pitch_conv_dims=9, pitch_rnn_dims=3, pitch_dropout=0.0, pitch_strength=0.7,
           energy_conv_dims=6, energy_rnn_dims=5, energy_dropout=0.0, energy_strength=1.0,
           rnn_dims=11, prenet_dims=10, prenet_k=5, postnet_num_highways=1,
           prenet_dropout=0.0, postnet_dims=9, postnet_k=2, prenet_num_highways=3,
           postnet_dropout=0.0, n_mels=7)

FULL = dict(embed_dims=256, series_embed_dims=64, num_chars=135,
            durpred_conv_dims=256, durpred_rnn_dims=64, durpred_dropout=0.5,
            pitch_conv_dims=256, pitch_rnn_dims=128, pitch_dropout=0.5, pitch_strength=1.0,
            energy_conv_dims=256, energy_rnn_dims=64, energy_dropout=0.5, energy_strength=1.0,
            rnn_dims=512, prenet_dims=256, prenet_k=16, postnet_num_highways=4,
            prenet_dropout=0.5, postnet_dims=256, postnet_k=8, prenet_num_highways=4,
            postnet_dropout=0.0, n_mels=80)

TRAIN_CFG = dict(dur_loss_factor=0.1, pitch_loss_factor=0.1, energy_loss_factor=0.1,
                 clip_grad_norm=1.0)


def load_npz(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def sub(d, prefix, as_torch=True):
    out = {}
    for k, v in d.items():
        if k.startswith(prefix):
            out[k[len(prefix):]] = torch.from_numpy(np.array(v)) if as_torch else v
    return out


def maxdiff(a, b):
    a = torch.as_tensor(a).detach().double()
    b = torch.as_tensor(b).detach().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.numel() == 0:
        return 0.0
    return float((a - b).abs().max())

TINY_MULTI = dict(embed_dims=16, series_embed_dims=8, num_chars=135,
                  durpred_conv_dims=16, durpred_rnn_dims=8, durpred_dropout=0.0,
                  pitch_conv_dims=16, pitch_rnn_dims=12, pitch_dropout=0.0, pitch_strength=1.0,
                  pitch_cond_conv_dims=12, pitch_cond_rnn_dims=8, pitch_cond_dropout=0.0,
                  energy_conv_dims=16, energy_rnn_dims=8, energy_dropout=0.0, energy_strength=0.5,
                  rnn_dims=20, prenet_dims=16, prenet_k=4, postnet_num_highways=2,
                  prenet_dropout=0.0, postnet_dims=12, postnet_k=3, prenet_num_highways=2,
                  postnet_dropout=0.0, n_mels=10, speaker_emb_dims=256, pitch_cond_emb_dims=4,
                  pitch_cond_categorical_dims=3)
TINY_FP = dict(num_chars=135,
               durpred_d_model=8, durpred_n_heads=2, durpred_layers=1, durpred_d_fft=12, durpred_dropout=0.0,
               pitch_d_model=8, pitch_n_heads=1, pitch_layers=2, pitch_d_fft=12, pitch_dropout=0.0,
               pitch_strength=1.0,
               energy_d_model=8, energy_n_heads=2, energy_layers=1, energy_d_fft=8, energy_dropout=0.0,
               energy_strength=0.5,
               d_model=16, conv1_kernel=5, conv2_kernel=1,
               prenet_layers=2, prenet_heads=2, prenet_fft=24, prenet_dropout=0.0,
               postnet_layers=2, postnet_heads=4, postnet_fft=20, postnet_dropout=0.0, n_mels=10)

FULL_FP = dict(num_chars=135,
               durpred_d_model=128, durpred_n_heads=2, durpred_layers=4, durpred_d_fft=128, durpred_dropout=0.5,
               pitch_d_model=128, pitch_n_heads=2, pitch_layers=4, pitch_d_fft=128, pitch_dropout=0.5,
               pitch_strength=1.0,
               energy_d_model=128, energy_n_heads=2, energy_layers=4, energy_d_fft=128, energy_dropout=0.5,
               energy_strength=1.0,
               d_model=256, conv1_kernel=9, conv2_kernel=1,
               prenet_layers=4, prenet_heads=2, prenet_fft=1024, prenet_dropout=0.1,
               postnet_layers=4, postnet_heads=2, postnet_fft=1024, postnet_dropout=0.1, n_mels=80)


TINY_MFP = dict(num_chars=135,
                durpred_d_model=8, durpred_n_heads=2, durpred_layers=1, durpred_d_fft=12, durpred_dropout=0.0,
                pitch_d_model=8, pitch_n_heads=1, pitch_layers=1, pitch_d_fft=12, pitch_dropout=0.0,
                pitch_strength=1.0,
                energy_d_model=8, energy_n_heads=2, energy_layers=1, energy_d_fft=8, energy_dropout=0.0,
                energy_strength=0.5,
                pitch_cond_d_model=8, pitch_cond_n_heads=2, pitch_cond_layers=1, pitch_cond_d_fft=8,
                pitch_cond_dropout=0.0, pitch_cond_output_dims=3,
                d_model=16, conv1_kernel=3, conv2_kernel=1,
                prenet_layers=1, prenet_heads=2, prenet_fft=24, prenet_dropout=0.0,
                postnet_layers=2, postnet_heads=4, postnet_fft=20, postnet_dropout=0.0, n_mels=10,
                speaker_emb_dims=8)


def sinusoid_pe(d_model, max_len=5000):
    """The reference's deterministic `pe` buffer (common_layers.py:134-141), [max_len,1,d]."""
    import math
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(1)


def fp_state(d, prefix):
    """FastPitch state_dict from a fixture: the truncated `pe` rows are checked against the formula and replaced
    by the full-length buffer."""
    sd = sub(d, prefix)
    for k in list(sd):
        if k.endswith('.pe'):
            full = sinusoid_pe(sd[k].shape[-1])
            assert maxdiff(full[:sd[k].shape[0]], sd[k]) == 0.0, k
            sd[k] = full
    return sd


TRAIN_CFG_MULTI = dict(dur_loss_factor=0.1, pitch_loss_factor=0.1, energy_loss_factor=0.1,
                       pitch_cond_loss_factor=0.1, clip_grad_norm=1.0)
