"""Hunt for a read of uninitialised memory in the MultiFastPitch train step: run the tiny golden step on a fresh
allocator, then again after filling + freeing a few hundred MB with a poison value (the caching allocator hands the
poisoned blocks back to the next torch.empty), and compare every parameter's gradient.   python lab/mfp_poison.py"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from helpers import TINY_MFP, TRAIN_CFG_MULTI, fp_state, load_npz, sub

Z = load_npz('tiny_multi_fastpitch.npz')


def run(poison):
    from forwardtacotron_amd.multi_fastpitch import MultiFastPitch
    from forwardtacotron_amd.trainer import TrainStep
    if poison is not None:
        blocks = [torch.full((n,), poison, device='cuda') for n in (1 << 10, 1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 24)
                  for _ in range(6)]
        torch.cuda.synchronize()
        del blocks
    m = MultiFastPitch(**TINY_MFP)
    m.load_state_dict(fp_state(Z, 'sd/'), strict=True)
    m = m.cuda()
    ts = TrainStep(m, lr=float(Z['lr']), train_cfg=TRAIN_CFG_MULTI)
    out = ts.step({k: v.clone().cuda() for k, v in sub(Z, 'batch/').items()})
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().clone().cpu() for n, p in m.named_parameters()}
    return float(out['loss']), float(out['grad_norm']), grads


ref = sub(Z, 'grad/')
for poison in (None, 3.0, float('nan'), 1e3):
    loss, gn, g = run(poison)
    print(f'poison {poison}: loss {loss:.6f} (golden {float(Z["loss/total"]):.6f})  grad_norm {gn:.6f} (golden {float(Z["grad_norm"]):.6f})')
    for k, v in g.items():
        if k in ref:
            r = torch.as_tensor(ref[k])
            d = (v - r).abs().max().item()
            if not d < 2e-5 + 2e-5 * r.abs().max().item():
                print('   differs:', k, tuple(v.shape), 'max diff', d, 'ref max', r.abs().max().item())
