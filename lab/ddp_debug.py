import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import torch.multiprocessing as mp
from test_gpu_parallel import _worker, _free_port, _model, _batch, LR
from helpers import TRAIN_CFG


def main():
    from forwardtacotron_amd.trainer import TrainStep
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 4096, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs: p.join(timeout=120)
    sds = [{k: torch.from_numpy(v) for k, v in r[1].items()} for r in res]
    grads = []
    for r in range(2):
        m = _model()
        ts = TrainStep(m, lr=0.0, train_cfg=TRAIN_CFG)
        ts.step({k: v.cuda() for k, v in _batch(r).items()})
        grads.append({n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()})
    m = _model()
    p0 = {n: p.detach().cpu().clone() for n, p in m.named_parameters()}

    def upd(g):
        norm = torch.sqrt(sum((v.double() ** 2).sum() for v in g.values())).float()
        coef = min(1.0, 1.0 / (float(norm) + 1e-6))
        out = {}
        for n in p0:
            gc = g[n] * coef
            out[n] = p0[n] - LR * (gc) / (gc.abs() + 1e-8 * (0.001 ** 0.5) / 1)  # sign-like first step
        return out, float(norm)
    mean = {n: 0.5 * (grads[0][n] + grads[1][n]) for n in p0}
    exp_mean, nm = upd(mean)
    exp_loc = [upd(grads[r]) for r in range(2)]
    print('norms: mean', nm, 'local', exp_loc[0][1], exp_loc[1][1], 'reported', res[0][2], res[1][2])
    for n in list(p0)[:400]:
        row = []
        for r in range(2):
            dm = float((sds[r][n] - exp_mean[n]).abs().max())
            dl = float((sds[r][n] - exp_loc[r][0][n]).abs().max())
            d0 = float((sds[r][n] - p0[n]).abs().max())
            row.append(f'r{r}: mean {dm:.1e} local {dl:.1e} moved {d0:.1e}')
        same = torch.equal(sds[0][n], sds[1][n])
        if not same:
            print(f'{n:45s} ' + ' | '.join(row))


if __name__ == '__main__':
    main()
