import sys, torch
sys.path.insert(0, '.')
from forwardtacotron_amd import hip as H
dev = 'cuda'
x = torch.randn(4096, 4096, device=dev); w = torch.randn(4096, 4096, device=dev)
for _ in range(3): H.linear_fwd(x, w)
torch.cuda.synchronize()
