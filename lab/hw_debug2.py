import os, sys, torch
sys.path.insert(0, '.')
from forwardtacotron_amd import ops
rows, C = int(sys.argv[1]), int(sys.argv[2])
g = torch.Generator().manual_seed(rows + C)
class Hw:
    def __init__(self):
        self.W1 = torch.nn.Linear(C, C); self.W2 = torch.nn.Linear(C, C)
        for p in (self.W1.weight, self.W2.weight): p.data = torch.randn(C, C, generator=g) * (1.0 / C ** 0.5)
        for p in (self.W1.bias, self.W2.bias): p.data = torch.randn(C, generator=g) * 0.3
        self.W1.cuda(); self.W2.cuda()
hs_all = [Hw() for _ in range(4)]
x = torch.randn(rows, C, generator=g); w = torch.randn(rows, C, generator=g)
for L in (1, 2, 3, 4):
    hs = hs_all[:L]
    res = {}
    for fused in ('1', '0'):
        os.environ['FT_HIGHWAY_FUSED'] = fused
        for h in hs:
            for p in (h.W1.weight, h.W1.bias, h.W2.weight, h.W2.bias): p.grad = None
        xg = x.cuda().requires_grad_(True)
        y = ops.highway_stack(xg, hs)
        (y * w.cuda()).sum().backward()
        res[fused] = (y.detach().cpu(), xg.grad.cpu().clone(), [p.grad.cpu().clone() for h in hs for p in (h.W1.weight, h.W1.bias, h.W2.weight, h.W2.bias)])
    print(L, 'y', float((res['1'][0] - res['0'][0]).abs().max()), 'dx', float((res['1'][1] - res['0'][1]).abs().max()),
          'grads', [round(float((a - b).abs().max()), 6) for a, b in zip(res['1'][2], res['0'][2])])
