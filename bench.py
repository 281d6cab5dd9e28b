"""bench.py -- mel-frames/s of one full ForwardTacotron train step on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = forward + 5 MaskedL1 losses + backward + clip_grad_norm(1.0) + Adam (+ bucketed RCCL all-reduce
when N > 1) on the LJSpeech-shaped synthetic batch of SURVEY.md section 8d (bs=32 per GPU, Tx=128,
Tm=841, 19,320 valid frames on rank 0's seed), singlespeaker.yaml model, fp32, dropout at config values,
inputs resident in HBM before the timed region.  Weak scaling: every rank draws its own bs=32 batch.

Rank 0 prints ONE JSON line (driver contract) with two extra objects:
  roofline     : the dominant GEMM launch (postnet conv bank forward) timed LIVE with a HIP event pair on its launch
                 stream inside every timed step -- algorithmic fp32 TFLOP/s against the f32 MFMA peak, the executed bf16
                 rate against the dense bf16 peak, HBM traffic per launch read from the committed PMC summary named in
                 `traffic_source` -- plus `families`: the weight-gradient GEMM in isolation, the TIME-dominant kernel family (the persistent recurrences,
                 latency-bound) and the LengthRegulator (HBM-bound), timed with event pairs in a short instrumented pass
                 AFTER the timed region (32 extra event records per step would perturb it), and whole-step TFLOP/s
  cpu_baseline : the reference step restated on stock fused torch CPU ops (oracle/ft_torch_cpu.py, pinned to the
                 reference goldens by tests/test_cpu_baseline.py) on the SAME bs=32 seed-0 batch, on this box's host cores
                 (rank 0, N=1 only), three whole steps (more if they fit the --cpu-budget) after a small warm-up
The event pairs are owned by this file: it wraps the package's launch wrappers (forwardtacotron_amd.hip.*) for the
duration of a measurement; the product carries no timing hook.
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
HBM_PEAK_TBS = 8.0                # HBM3E spec (6.3 TB/s achievable per the guide)

BANK_SHAPE = (32, 841, 80, 256, 8)      # B, T, Cin, C, K of the postnet conv bank forward at the benchmark config
TRAFFIC_FILE = os.path.join('profiles', 'r03_pmc_bank_fwd.json')    # written by tools/pmc_traffic.py from --pmc passes


class Probes:
    """HIP event pairs around selected launch wrappers of forwardtacotron_amd.hip, recorded on the stream the wrapper
    launches on (torch's current stream at call time)."""

    def __init__(self, hip_mod):
        self.hip, self.saved, self.records = hip_mod, {}, []

    def wrap(self, name, describe):
        """describe(args, kwargs, result) -> dict (or None to skip the call)"""
        orig = getattr(self.hip, name)
        self.saved[name] = orig

        def timed(*a, **k):
            info = describe(a, k)
            if info is None:
                return orig(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(*a, **k)
            e1.record()
            self.records.append((name, info, e0, e1))
            return r

        setattr(self.hip, name, timed)

    def remove(self):
        for n, f in self.saved.items():
            setattr(self.hip, n, f)
        self.saved = {}

    def results(self):
        return [(n, info, e0.elapsed_time(e1)) for n, info, e0, e1 in self.records]


def _bank_describe(a, k):
    x, K, C = a[0], a[2], a[3]                  # hip.conv_bank_fwd_stats(x, wp_all, K, C, relu)
    return {} if tuple(x.shape) + (C, K) == BANK_SHAPE else None


def dominant_kernel_roofline(bank_ms):
    """The dominant GEMM-shaped launch of the step -- the postnet conv bank forward, one ft_gemm_rows_b3p_kernel<3>
    launch (M = 32*842 rows, 8 members k=1..8, Cin 80 -> 256: 2*B*(T+1)*80*256*36 FLOP)."""
    B, T, Cin, C, K = BANK_SHAPE
    ms = sum(bank_ms) / max(len(bank_ms), 1)
    flops = 2.0 * B * (T + 1) * Cin * C * (K * (K + 1) // 2)
    ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    b3 = os.environ.get('FT_GEMM_B3', '1') != '0'
    piped = os.environ.get('FT_GEMM_PIPE', '1') != '0'
    kernel = ('ft_gemm_rows_b3p_kernel<3>' if piped else 'ft_gemm_rows_b3_kernel<2,2>') if b3 else 'ft_gemm_rows_kernel<2,2,NT>'
    out = {'bound': 'mfma', 'kernel': kernel + ' (postnet conv bank fwd)',
           'achieved': round(ach, 2), 'peak': F32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
           'frac': round(ach / F32_MFMA_PEAK_TFLOPS, 4),
           'launch_ms': round(ms, 4), 'launches_timed': len(bank_ms), 'flops_per_launch': flops,
           'algorithmic_bytes': 4.0 * (B * (T + 1) * Cin + K * (K + 1) // 2 * Cin * C + B * (T + 1) * K * C),
           'traffic': None, 'traffic_unit': 'B/launch', 'traffic_source': None}
    # HBM-side bytes per launch: NOT measured in this run (PMC collection needs its own rocprofv3 passes); taken from
    # the committed summary of those passes if it describes this kernel, else left null
    try:
        t = json.load(open(os.path.join(ROOT, TRAFFIC_FILE)))
        if t.get('kernel', '').startswith(kernel.split('<')[0]) and tuple(t.get('shape', ())) == BANK_SHAPE:
            out['traffic'] = t['traffic_bytes_per_launch']
            out['traffic_source'] = TRAFFIC_FILE + ' (offline rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 ' \
                                                   'read correction)'
    except (OSError, ValueError, KeyError):
        pass
    if b3:
        # fp32 work executed on the bf16 matrix pipe: every fp32 product = 6 bf16 MFMA products of an exact 3-way
        # operand split, fp32 accumulation.  `achieved` stays the ALGORITHMIC fp32 rate (priced against the fp32 MFMA
        # peak); the executed bf16 rate and its share of the dense bf16 peak are stated next to it.
        out['pipe'] = 'bf16 MFMA x6 per fp32 product (exact 3-way split, fp32 accumulate)'
        out['executed_bf16_tflops'] = round(6 * ach, 1)
        out['frac_of_bf16_dense_peak'] = round(6 * ach / BF16_MFMA_PEAK_TFLOPS, 4)
    return out


def family_rooflines(recs, steps, batch, step_ms, lr_iso_ms=None):
    """Per-step totals of the recurrence launches (time-dominant family) and of the LengthRegulator expansion."""
    B = int(batch['x'].shape[0])
    trunk = {'ms': 0.0, 'flop': 0.0, 'dep_steps': 0, 'launches': 0}
    side = {'ms': 0.0, 'flop': 0.0, 'dep_steps': 0, 'launches': 0}
    lr_ms, lr_bytes = 0.0, 0.0
    for name, info, ms in recs:
        if name in ('lr_expand', 'lr_expand_tm'):
            lr_ms += ms
            lr_bytes += info['bytes']
            continue
        acc = side if info['T'] <= int(batch['x'].shape[1]) and info['H'] <= 128 else trunk
        acc['ms'] += ms
        acc['flop'] += info['flop']
        acc['dep_steps'] += info['T']
        acc['launches'] += 1
    out = []
    for tag, acc in (('trunk (prenet GRU-256 over 128 tokens, LSTM-512 and postnet GRU-256 over 841 frames; the step\'s critical stream)', trunk),
                     ('predictors (3 x GRU, side stream, overlapped; the event pairs include the device-side waits behind the trunk\'s launches)', side)):
        if not acc['launches']:
            continue
        ms, tf = acc['ms'] / steps, acc['flop'] / steps / (acc['ms'] / steps * 1e-3) / 1e12
        out.append({'family': 'persistent recurrences, ' + tag, 'bound': 'latency (cross-CU hand-off per dependent step)',
                    'kernels': 'ft_rnn_fwd_persist_kernel / ft_rnn_bwd_persist_kernel / ft_rnn_bwd_rs_kernel',
                    'ms_per_step': round(ms, 3), 'share_of_step': round(ms / step_ms, 3),
                    'launches_per_step': acc['launches'] // steps, 'dependent_steps': acc['dep_steps'] // steps,
                    'us_per_dependent_step': round(ms * 1e3 / (acc['dep_steps'] / steps), 3),
                    'achieved': round(tf, 2), 'peak': F32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(tf / F32_MFMA_PEAK_TFLOPS, 4)})
    if lr_ms > 0:
        ms = lr_iso_ms if lr_iso_ms else lr_ms / steps
        tbs = lr_bytes / steps / (ms * 1e-3) / 1e12
        out.append({'family': 'LengthRegulator expand (ft_lr_expand_kernel; in the regulated-LSTM path it carries the '
                              'per-token input projection, 4096 floats per row, into the recurrence\'s time-major layout)',
                    'bound': 'hbm',
                    'launch_ms': round(ms, 4), 'timing': '50 launches back to back between one event pair' if lr_iso_ms
                    else 'event pair around the in-step launch', 'in_step_event_pair_ms': round(lr_ms / steps, 4),
                    'bytes_per_launch': lr_bytes / steps,
                    'achieved': round(tbs * 1e3, 1), 'peak': HBM_PEAK_TBS * 1e3, 'unit': 'GB/s',
                    'frac': round(tbs / HBM_PEAK_TBS, 4)})
    return out


def install_family_probes(probes, batch):
    lens_sum = float(batch['mel_len'].sum())

    def rnn(G, has_lens):
        def d(a, k):
            # gru_fwd(xp, ..., H, save) / gru_bwd(dout, out, gates, whhT_f, whhT_r, H) / lstm_fwd(xp, ..., lens, H, save)
            # / lstm_bwd(dout, raw, cst, gates, whhT_f, whhT_r, lens, H): first tensor is time-major [T,B,*]
            T, Bq = int(a[0].shape[0]), int(a[0].shape[1])
            H = [v for v in a if isinstance(v, int)][0]
            lens = next((v for v in a if torch.is_tensor(v) and v.dtype == torch.int64), None) if has_lens else None
            valid = lens_sum if lens is not None else float(Bq * T)
            return {'T': T, 'H': H, 'flop': 2.0 * valid * 2 * G * H * H}
        return d

    probes.wrap('gru_fwd', rnn(3, False))
    probes.wrap('gru_bwd', rnn(3, False))
    probes.wrap('lstm_fwd', rnn(4, True))
    probes.wrap('lstm_bwd', rnn(4, True))

    def lr(a, k):
        x, cum, Tm = a[0], a[1], a[2]
        Bq, Tx, C = x.shape
        probes.lr_args = (x, cum, Tm)            # for the back-to-back timing below (the operands of the last step)
        probes.lr_fn = 'lr_expand_tm' if len(a) > 3 or k else 'lr_expand'
        # algorithmic bytes: one 4*C-byte row read per valid token + one written per output frame (B x Tm)
        return {'bytes': 4.0 * C * (float(batch['x_len'].sum()) + Bq * Tm)}

    probes.wrap('lr_expand', lr)
    probes.wrap('lr_expand_tm', lr)          # (x, cum, Tm, pad_row, want_src=...)


def lr_back_to_back_ms(hip_mod, args, n=50, fn='lr_expand'):
    """An event pair around ONE 12-us launch mostly measures the events; n launches back to back between one pair give the
    kernel's own time (+ the ~1.5 us dependent-launch boundary).  The output tensor is re-allocated per call by the
    wrapper (caching allocator: no device work)."""
    x, cum, Tm = args
    expand = getattr(hip_mod, fn)
    for _ in range(3):
        expand(x, cum, Tm)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        expand(x, cum, Tm)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def wgrad_family(hip_mod, device, n=20):
    """The weight-gradient GEMM of the step with the most FLOPs per launch -- the decoder LSTM's W_ih gradient of one
    direction: dW[2048,512] = dgates[26912 rows, time-major, 2048 of 4096 columns]^T x[26912,512] -- launched n times back
    to back between one event pair (in the step these launches run on the side stream beside other work; their in-step
    durations are contended).  One launch = ft_gemm_tn_b3p_kernel (4 row splits) + the ordered slab sum."""
    B, T, I, GH = 32, 841, 512, 2048
    g = torch.Generator(device='cpu').manual_seed(3)
    dy = torch.randn(T, B, 2 * GH, generator=g).to(device)
    x = torch.randn(B, T, I, generator=g).to(device)
    dw = torch.empty(GH, I, device=device)

    def launch():
        hip_mod.linear_bwd_weight_raw(dy.data_ptr(), 2 * GH, x.data_ptr(), I, dw, B * T, I, GH, B=B, T=T, dy_tm=True,
                                      x_tm=False)
    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    flops = 2.0 * B * T * I * GH
    ach = flops / (ms * 1e-3) / 1e12
    b3 = os.environ.get('FT_GEMM_B3', '1') != '0'
    piped = os.environ.get('FT_GEMM_TN_PIPE', '1') != '0'
    rec = {'family': 'weight-gradient GEMM (decoder LSTM W_hh, one direction: 2048 x 512 over 26912 time-major rows)',
           'bound': 'mfma', 'kernels': ('ft_gemm_tn_b3p_kernel' if piped else 'ft_gemm_tn_b3_kernel') if b3 else 'ft_gemm_tn_kernel',
           'launch_ms': round(ms, 4), 'timing': f'{n} launches back to back between one event pair (incl. the ordered slab sum)',
           'flops_per_launch': flops, 'achieved': round(ach, 2), 'peak': F32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
           'frac': round(ach / F32_MFMA_PEAK_TFLOPS, 4)}
    if b3:
        rec['pipe'] = 'bf16 MFMA x6 per fp32 product (exact 3-way split, fp32 accumulate)'
        rec['executed_bf16_tflops'] = round(6 * ach, 1)
        rec['frac_of_bf16_dense_peak'] = round(6 * ach / BF16_MFMA_PEAK_TFLOPS, 4)
    return rec


def cpu_baseline(model_cfg, train_cfg, budget_s, threads):
    """The reference step on stock fused torch CPU ops (oracle/ft_torch_cpu.py), SAME bs=32 seed-0 batch as the GPU
    run: a B=2 warm-up step (thread pools, allocator), then at least THREE whole bs=32 steps (more while they fit the budget).  Threads: the step is ~3,650 dependent recurrence timesteps of small matmuls plus their autograd, which does
    not scale with cores -- measured on the MI355X host (256 cpus): 128 torch threads 226 s/step, 8 threads (build
    container) 51 s/step; the default is 16 threads = one GPU's share of the host, stated in `cores`."""
    torch.set_num_threads(max(1, min(threads, os.cpu_count() or 1)))
    from oracle import ft_torch_cpu as C        # baseline leg only
    from forwardtacotron_amd import data
    from forwardtacotron_amd.model import ForwardTacotron
    torch.manual_seed(0)
    m = ForwardTacotron(**model_cfg)
    P = {k: v.clone() for k, v in m.state_dict().items()}
    tr = C.CpuTrainer(P, model_cfg, train_cfg, lr=5e-5)
    tr.step(data.synthetic_batch(B=2, Tmax=128, n_mels=model_cfg['n_mels'], seed=1))
    batch = data.synthetic_batch(B=32, Tmax=128, n_mels=model_cfg['n_mels'], seed=0)
    n_frm = int(batch['mel_len'].sum())
    times = []
    t_all = time.time()
    # >= 3 timed steps after the warm-up, whatever they cost (~45 s each on 16 threads); the budget only decides
    # about a fourth and fifth
    while len(times) < 3 or (len(times) < 5 and time.time() - t_all + max(times) < budget_s):
        t0 = time.time()
        tr.step(batch)
        times.append(time.time() - t0)
    dt = sum(times) / len(times)
    return {'value': round(n_frm / dt, 1), 'unit': 'frames/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{len(times)} full train step(s) of the SAME bs=32 seed-0 batch ({n_frm} frames, '
                      f'{dt:.1f} s/step) after a B=2 warm-up step; oracle/ft_torch_cpu.py (stock fused torch CPU ops: '
                      f'conv1d, batch_norm, _VF.gru/lstm packed, autograd, Adam), torch threads '
                      f'{torch.get_num_threads()}, nproc {os.cpu_count()}',
            'step_seconds': [round(t, 2) for t in times]}


def fastpitch_bf16_record(steps=10, warmup=3):
    """BASELINE configs[2] beside the headline line: the FastPitch train step at bs=32 / Tx=128 / Tm=841 with bf16 matmuls
    (operands rounded to bf16, one bf16 MFMA per product, fp32 accumulation; LayerNorm / softmax / losses / Adam fp32),
    priced against the dense bf16 MFMA peak (algorithmic FLOPs: SURVEY.md section 8d, 94.6 MFLOP per valid frame).
    Measured by tools/bench_variants.py in a CHILD process: in this process, right behind the ForwardTacotron
    measurement, the same step ran 1.7-2x slower (31-37 ms against 18.4 ms alone; the caching allocator's block pool is
    shaped by the first model) -- a number about this process, not about the kernels."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, 'tools', 'bench_variants.py'), '--model', 'fastpitch', '--mode', 'train',
           '--dtype', 'bf16', '--steps', str(steps), '--warmup', str(warmup)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    except Exception as e:                      # the headline line must not depend on the extra record
        return {'error': f'{type(e).__name__}: {e}'[:300]}
    return {'workload': 'FastPitch singlespeaker.yaml train step, bs=32, Tx=128, Tm=841, bf16 matmuls (BASELINE configs[2])',
            'dtype': 'bf16', 'ms_per_step': rec['ms_per_step'], 'frames_per_s': rec['frames_per_s'], 'steps': steps,
            'algorithmic_tflops': rec['algorithmic_tflops'], 'frac_of_bf16_dense_peak': rec['frac_of_bf16_dense_peak'],
            'loss': rec.get('loss'), 'measured_by': 'tools/bench_variants.py (child process)',
            'parity': 'defined by this repo against the fp32 oracle (tests/test_gpu_fastpitch.py); the reference has no '
                      'bf16 path'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--family-steps', type=int, default=3, help='instrumented steps after the timed region (rank 0)')
    ap.add_argument('--cpu-budget', type=float, default=150.0,
                    help='seconds of CPU baseline work beyond which no 4th / 5th step is started (3 are always timed)')
    ap.add_argument('--cpu-threads', type=int, default=16, help='torch threads of the CPU baseline')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-variants', action='store_true', help='skip the FastPitch bf16 record (N=1 only)')
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
            # a collective's workgroups hold CUs the persistent recurrences' grids may be waiting for (they retire on
            # their own -- a 24 MB bucket is < 1 ms against a spin bound of >= 100 ms -- so this is about delay, not
            # faults): at most 16 of the 256 CUs, which still drives all seven xGMI links
            os.environ.setdefault('NCCL_MAX_NCHANNELS', '16')
            dist.init_process_group('nccl', device_id=device)

    from forwardtacotron_amd import data
    from forwardtacotron_amd.model import ForwardTacotron
    from forwardtacotron_amd.trainer import TrainStep
    from forwardtacotron_amd import _lib as _ftlib, hip as _hip

    model_cfg = dict(data.SINGLESPEAKER_MODEL)
    train_cfg = dict(data.SINGLESPEAKER_TRAIN)
    batch = data.to_device(data.synthetic_batch(B=32, Tmax=128, n_mels=80, seed=rank), device)
    n_frm = int(batch['mel_len'].sum())
    n_tok = int(batch['x_len'].sum())
    dur0 = batch['dur'].clone()
    if rehearsal and world > 1:
        # several ranks share one GPU here: their persistent grids cannot all be co-resident
        _ftlib.query('ft_rnn_set_persistent', 0)

    def build():
        torch.manual_seed(0)                  # identical initial weights on every rank
        model = ForwardTacotron(**model_cfg).to(device)
        return TrainStep(model, lr=5e-5, train_cfg=train_cfg, gc_freeze=True)   # (process-wide; the application's call)

    def measure(ts):
        """W untimed warm-up steps, then EXACTLY K timed steps between barrier + synchronize."""
        def one_step():
            batch['dur'].copy_(dur0)          # the LengthRegulator clamps dur in place
            return ts.step(batch)

        for _ in range(args.warmup):
            one_step()
        probes = Probes(_hip)
        if rank == 0:
            probes.wrap('conv_bank_fwd_stats', _bank_describe)      # two event records per step
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
        probes.remove()
        bank_ms = [ms for _, _, ms in probes.results()]
        fam = []
        try:
            _hip.check_rnn_status()           # raises if a persistent recurrence hit its spin bound (sticky word)
            ok = 1.0
        except _ftlib.FtError:
            ok = 0.0
        if ok and world == 1 and args.family_steps > 0:      # (extra steps on one rank only would hang the all-reduce)
            fp = Probes(_hip)
            install_family_probes(fp, batch)
            for _ in range(args.family_steps):
                one_step()
            torch.cuda.synchronize()
            fp.remove()
            lr_iso = (lr_back_to_back_ms(_hip, fp.lr_args, fn=getattr(fp, 'lr_fn', 'lr_expand'))
                      if getattr(fp, 'lr_args', None) else None)
            fam = family_rooflines(fp.results(), args.family_steps, batch, elapsed / args.steps * 1e3, lr_iso)
            fam.append(wgrad_family(_hip, device))
        flag = torch.tensor([ok], device=device)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return elapsed, o, bool(flag.item() > 0), bank_ms, fam

    dt, out, rnn_ok, bank_ms, fam = measure(build())
    rnn_persistent = not (rehearsal and world > 1)
    if not rnn_ok:
        # a persistent recurrence timed out on some rank: the device skipped every update from then on, so the run
        # timed something else -- switch every rank to the per-timestep kernels, rebuild model and trainer from the
        # seed and measure again
        _ftlib.query('ft_rnn_set_persistent', 0)
        rnn_persistent = False
        dt, out, rnn_ok, bank_ms, fam = measure(build())
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
        roof = dominant_kernel_roofline(bank_ms)
        roof['families'] = fam
        roof['step'] = {'algorithmic_tflops': round(step_tflops, 2),
                        'frac_of_f32_mfma_peak': round(step_tflops / (F32_MFMA_PEAK_TFLOPS * world), 4),
                        'frac_of_bf16_dense_peak': round(step_tflops / (BF16_MFMA_PEAK_TFLOPS * world), 4),
                        'flop_per_valid_frame': 64.5e6}
        pers, refused = _hip.rnn_counters()
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
            'rnn_launches': {'persistent': pers, 'per_step_fallback': refused, 'waited_for_other_stream': _hip.rnn_waited_launches()},
            # (direction, batch group) groups of the persistent launches since load: hand-off through their XCD's L2 /
            # on the agent-scope protocol (placement is verified inside each kernel, ft_rnn_mode_counts)
            'rnn_modes': dict(zip(('xcd_local_groups', 'agent_scope_groups'), _hip.rnn_mode_counts())),
            'roofline': roof,
        }
        if world == 1 and not args.no_variants:
            line['variants'] = {'fastpitch_bf16_train': fastpitch_bf16_record()}
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(model_cfg, train_cfg, args.cpu_budget, args.cpu_threads)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
