"""Secondary measurements (NOT the driver contract -- that is /bench.py): train-step and inference throughput of
the other §8 variants on one MI355X, one JSON line per run.

    python tools/bench_variants.py --model fastpitch --mode train  [--steps 10 --warmup 3 --batch 32]
    python tools/bench_variants.py --model multi     --mode train  --batch 64          # BASELINE configs[3]
    python tools/bench_variants.py --model forward   --mode infer  --batch 128 --tokens 1000   # configs[4], long-form
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build(name, device):
    from forwardtacotron_amd import data
    if name == 'fastpitch':
        from forwardtacotron_amd.fastpitch import FastPitch
        return FastPitch(**data.FASTPITCH_MODEL).to(device), data.FASTPITCH_MODEL
    if name == 'multi':
        from forwardtacotron_amd.multi_model import MultiForwardTacotron
        return MultiForwardTacotron(**data.MULTISPEAKER_MODEL).to(device), data.MULTISPEAKER_MODEL
    from forwardtacotron_amd.model import ForwardTacotron
    return ForwardTacotron(**data.SINGLESPEAKER_MODEL).to(device), data.SINGLESPEAKER_MODEL


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', choices=['forward', 'fastpitch', 'multi'], default='fastpitch')
    ap.add_argument('--mode', choices=['train', 'infer'], default='train')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--tokens', type=int, default=128)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-dropout', action='store_true')
    ap.add_argument('--dtype', choices=['fp32', 'bf16'], default='fp32',
                    help="bf16: matmul operands rounded to bf16, fp32 accumulation (FastPitch variants; BASELINE configs[2])")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit('needs an MI355X: the hot path has no CPU fallback')
    device = torch.device('cuda', 0)
    from forwardtacotron_amd import data, hip
    from forwardtacotron_amd.trainer import TrainStep
    torch.manual_seed(0)
    model, cfg = build(args.model, device)
    if args.dtype == 'bf16':
        if not hasattr(model, 'matmul_dtype'):
            raise SystemExit('--dtype bf16 applies to the FastPitch variants (recurrent models stay fp32)')
        model.matmul_dtype = 'bf16'
    if args.no_dropout:
        for m in model.modules():
            if hasattr(m, 'p'):
                m.p = 0.0
    batch = data.synthetic_batch(B=args.batch, Tmax=args.tokens, n_mels=80, seed=0)
    if args.model == 'multi':
        batch['pitch_cond'] = ((batch['pitch'] != 0).long() + 1) * (batch['x'] > 0).long()
        se = torch.randn(args.batch, 256, generator=torch.Generator().manual_seed(1))
        batch['speaker_emb'] = se / se.norm(dim=1, keepdim=True)
    batch = data.to_device(batch, device)
    n_frm, n_tok = int(batch['mel_len'].sum()), int(batch['x_len'].sum())
    Tm = int(batch['mel_len'].max())
    dur0 = batch['dur'].clone()

    if args.mode == 'train':
        ts = TrainStep(model, lr=5e-5, train_cfg=dict(data.SINGLESPEAKER_TRAIN, pitch_cond_loss_factor=0.1), gc_freeze=True)

        def step():
            batch['dur'].copy_(dur0)
            return ts.step(batch)
    else:
        model.eval()

        def step():
            # the mel-generation path with the batch's own durations (an untrained duration predictor degenerates
            # to the fill_(2.) fallback -- SURVEY.md section 8d, config 5)
            batch['dur'].copy_(dur0)
            with torch.no_grad():
                return model(batch)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    c0 = hip.rnn_counters()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    hip.check_rnn_status()
    if args.model == 'fastpitch':
        fl = data.fastpitch_train_flops(n_tok, n_frm, args.tokens, Tm)
    elif args.model == 'multi':
        fl = 3.0 * 2.0 * (19_734_656 * n_tok + 9_068_544 * n_frm)
    else:
        fl = data.train_flops(n_tok, n_frm)
    if args.mode == 'infer':
        fl /= 3.0
    line = {'model': args.model, 'mode': args.mode, 'batch': args.batch, 'Tx': args.tokens, 'Tm': Tm,
            'frames': n_frm, 'tokens': n_tok, 'ms_per_step': round(dt * 1e3, 3),
            'frames_per_s': round(n_frm / dt, 1), 'algorithmic_tflops': round(fl / dt / 1e12, 2),
            'frac_of_f32_mfma_peak': round(fl / dt / 157.3e12, 4),
            'frac_of_bf16_dense_peak': round(fl / dt / 2500e12, 4), 'dtype': 'f32' if args.dtype == 'fp32' else 'bf16',
            'peak_mem_GiB': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}
    c1 = hip.rnn_counters()
    # recurrence launches per step in the timed region: persistent ones / launches refused the persistent form (each
    # refusal = T per-step launches, unless the forward then ran as persistent launches over batch slices)
    line['rnn_launches_per_step'] = {'persistent': (c1[0] - c0[0]) / args.steps, 'refused': (c1[1] - c0[1]) / args.steps}
    if args.mode == 'train':
        line['loss'] = round(float(out['loss']), 5)
    print(json.dumps(line), flush=True)


if __name__ == '__main__':
    main()
