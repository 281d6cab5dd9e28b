"""bench.py -- mel-frames/s of one full ForwardTacotron train step on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = forward + 5 MaskedL1 losses + backward + clip_grad_norm(1.0) + Adam (+ bucketed RCCL all-reduce
when N > 1) on the LJSpeech-shaped synthetic batch of SURVEY.md section 8d (bs=32 per GPU, Tx=128,
Tm=841, 19,320 valid frames on rank 0's seed), singlespeaker.yaml model, fp32, dropout at config values,
inputs resident in HBM before the timed region.  Weak scaling: every rank draws its own bs=32 batch.

Rank 0 prints ONE JSON line (driver contract) with two extra objects:
  roofline     : MFMA roofline of the dominant GEMM launch (postnet conv bank forward), timed live with HIP events
                 on its launch stream inside every timed step: algorithmic fp32 TFLOP/s against the f32 MFMA peak,
                 and -- the kernel runs fp32 products as exact 3-way bf16 splits on the bf16 matrix pipe -- the
                 executed bf16 rate against the dense bf16 peak; plus whole-step achieved TFLOP/s in `step`
  cpu_baseline : the CPU oracle (port of the reference step) timed on this box's host cores on a bounded
                 sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA peak (same guide; AMD's 5 PF headline includes 2:1 sparsity)


BANK_SHAPE = (32, 841, 80, 256, 8)      # B, T, Cin, C, K of the postnet conv bank forward at the benchmark config


def dominant_kernel_roofline(events):
    """The dominant GEMM-shaped launch of the step -- the postnet conv bank forward, one ft_gemm_rows_kernel<2,2,NT>
    launch (M = 32*842 rows, 8 members k=1..8, Cin 80 -> 256: 2*B*(T+1)*80*256*36 FLOP) -- timed LIVE: HIP events
    recorded around that launch on its own stream inside every timed step (forwardtacotron_amd.hip.bank_probe)."""
    B, T, Cin, C, K = BANK_SHAPE
    ms = sum(a.elapsed_time(b) for a, b in events) / max(len(events), 1)
    flops = 2.0 * B * (T + 1) * Cin * C * (K * (K + 1) // 2)
    ach = flops / (ms * 1e-3) / 1e12
    b3 = os.environ.get('FT_GEMM_B3', '1') != '0'
    out = {'bound': 'mfma',
           'kernel': ('ft_gemm_rows_b3_kernel<2,2>' if b3 else 'ft_gemm_rows_kernel<2,2,NT>') + ' (postnet conv bank fwd)',
           'achieved': round(ach, 2), 'peak': F32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
           'frac': round(ach / F32_MFMA_PEAK_TFLOPS, 4),
           # HBM-side bytes per launch from the PMC counters (profiles/r01e_pmc_bank_fwd.txt: FETCH_SIZE x 2 (gfx950
           # correction for 16-B-per-lane reads) + WRITE_SIZE, separate rocprofv3 --pmc passes); algorithmic bytes 232 MB
           'traffic': 401.6e6, 'traffic_unit': 'B/launch', 'algorithmic_bytes': 232.2e6,
           'launch_ms': round(ms, 4), 'launches_timed': len(events), 'flops_per_launch': flops}
    if b3:
        # fp32 work executed on the bf16 matrix pipe: every fp32 product = 6 bf16 MFMA products of an exact 3-way
        # operand split, fp32 accumulation.  `achieved` stays the ALGORITHMIC fp32 rate (priced against the fp32 MFMA
        # peak); the executed bf16 rate and its share of the dense bf16 peak are stated next to it.
        out['pipe'] = 'bf16 MFMA x6 per fp32 product (exact 3-way split, fp32 accumulate)'
        out['executed_bf16_tflops'] = round(6 * ach, 1)
        out['frac_of_bf16_dense_peak'] = round(6 * ach / BF16_MFMA_PEAK_TFLOPS, 4)
    return out


def cpu_baseline(model_cfg, train_cfg):
    """Oracle (CPU port of the reference train step) on a bounded sample: B=2 items of the same shape."""
    from oracle import ft_oracle as O       # checker / baseline leg only
    from forwardtacotron_amd.model import ForwardTacotron
    torch.manual_seed(0)
    m = ForwardTacotron(**model_cfg)
    P = {k: v.clone() for k, v in m.state_dict().items()}
    batch = O.synthetic_batch(B=2, Tmax=128, n_mels=model_cfg['n_mels'], seed=0)
    n_frm = int(batch['mel_len'].sum())
    t0 = time.time()
    O.train_step(P, {}, batch, model_cfg, train_cfg, lr=5e-5, step_count=1)
    dt = time.time() - t0
    return {'value': round(n_frm / dt, 1), 'unit': 'frames/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'1 train step, B=2 of the bs=32 workload ({n_frm} frames, {dt:.1f} s), '
                      f'oracle/ft_oracle.py on {os.cpu_count()} host cpus'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the hot path has no CPU fallback')
    # Rehearsal hook for a one-GPU box (never set by the driver): FT_BENCH_REHEARSAL=1 puts every rank on cuda:0 and
    # talks over gloo, so that the N > 1 code path (barriers, bucketed all-reduce, rank-0 reporting) can be exercised
    # where RCCL cannot be (it refuses two ranks on one device).  The numbers of such a run mean nothing.
    rehearsal = os.environ.get('FT_BENCH_REHEARSAL') == '1'
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if rehearsal:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=device)

    from forwardtacotron_amd import data
    from forwardtacotron_amd.model import ForwardTacotron
    from forwardtacotron_amd.trainer import TrainStep

    model_cfg = dict(data.SINGLESPEAKER_MODEL)
    train_cfg = dict(data.SINGLESPEAKER_TRAIN)
    torch.manual_seed(0)                      # identical initial weights on every rank
    model = ForwardTacotron(**model_cfg).to(device)
    ts = TrainStep(model, lr=5e-5, train_cfg=train_cfg)
    batch = data.to_device(data.synthetic_batch(B=32, Tmax=128, n_mels=80, seed=rank), device)
    n_frm = int(batch['mel_len'].sum())
    n_tok = int(batch['x_len'].sum())
    dur0 = batch['dur'].clone()

    def one_step():
        batch['dur'].copy_(dur0)              # the LengthRegulator clamps dur in place
        return ts.step(batch)

    from forwardtacotron_amd import _lib as _ftlib, hip as _hip
    probe = {'shape': BANK_SHAPE, 'events': []}
    if rehearsal and world > 1:
        # several ranks share one GPU here: their persistent grids cannot all be co-resident
        _ftlib.query('ft_rnn_set_persistent', 0)

    def measure():
        """W untimed warm-up steps, then EXACTLY K timed steps between barrier + synchronize."""
        for _ in range(args.warmup):
            one_step()
        probe['events'] = []
        if rank == 0:
            _hip.bank_probe = probe           # two event records per step: no synchronisation, no extra launches
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            o = one_step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        _hip.bank_probe = None
        try:
            _hip.check_rnn_status()           # raises if a persistent recurrence hit its spin bound
            ok = 1.0
        except _ftlib.FtError:
            ok = 0.0
        flag = torch.tensor([ok], device=device)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return elapsed, o, bool(flag.item() > 0)

    dt, out, rnn_ok = measure()
    rnn_persistent = not (rehearsal and world > 1)
    if not rnn_ok:
        # a persistent recurrence timed out on some rank (its workgroups were not co-resident, e.g. under heavy
        # contention): that run is invalid -- switch every rank to the per-timestep kernels and measure again
        _ftlib.query('ft_rnn_set_persistent', 0)
        rnn_persistent = False
        dt, out, rnn_ok = measure()
        if not rnn_ok:
            raise SystemExit('bench.py: recurrence status still bad with the per-step kernels')
    stats = torch.tensor([dt, float(n_frm), float(n_tok)], device=device, dtype=torch.float64)
    if world > 1:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt = float(mx[0])
        tot_frm, tot_tok = float(sm[1]), float(sm[2])
    else:
        tot_frm, tot_tok = float(n_frm), float(n_tok)
    loss = float(out['loss'])

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = tot_frm * args.steps / dt
        step_tflops = data.train_flops(tot_tok, tot_frm) / (dt / args.steps) / 1e12
        roof = dominant_kernel_roofline(probe['events'])
        roof['step'] = {'algorithmic_tflops': round(step_tflops, 2),
                        'frac_of_f32_mfma_peak': round(step_tflops / (F32_MFMA_PEAK_TFLOPS * world), 4),
                        'flop_per_valid_frame': 64.5e6}
        line = {
            'metric': 'mel_frames_per_sec_train_step', 'value': round(value, 1), 'unit': 'frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': 'LJSpeech singlespeaker.yaml ForwardTacotron train step, bs=32/GPU, Tx=128, '
                                   'Tm=841, fp32 (BASELINE configs[1])',
                       'global_batch': 32 * world, 'frames_per_step': tot_frm, 'parallelism': f'dp{world}',
                       'arithmetic': 'fp32 tensors and accumulation everywhere; recurrences and small GEMMs on f32 '
                                     'MFMA, 128x128-tile GEMMs as exact 3-way bf16 splits on bf16 MFMA (fp32-accurate: '
                                     'same parity bars, FT_GEMM_B3=0 switches it off)'},
            'per_gpu': round(value / world, 1), 'loss': round(loss, 5), 'rnn_persistent': rnn_persistent,
            'roofline': roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(model_cfg, train_cfg)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
