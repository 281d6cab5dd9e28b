import math, os, sys, torch
sys.path.insert(0, '.')
from forwardtacotron_amd import hip as H
from forwardtacotron_amd import fastpitch as FP
torch.manual_seed(0)
B, T, d, nh = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 841, 256, 2
hd = d // nh
qkv = torch.randn(B, T, 3 * d, device='cuda'); datt = torch.randn(B, T, d, device='cuda')
lens = torch.randint(T // 2, T + 1, (B,)); lens[0] = T
kp = (torch.arange(T)[None, :] >= lens[:, None]).to(torch.uint8).cuda()
def t(f, n=5):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for p in (0.0, 0.1):
    att, lse = H.attn_fwd(qkv, kp, nh, 1 / math.sqrt(hd), p, 7)
    tf = t(lambda: H.attn_fwd(qkv, kp, nh, 1 / math.sqrt(hd), p, 7))
    tb = t(lambda: H.attn_bwd(qkv, att, datt, kp, lse, nh, 1 / math.sqrt(hd), p, 7))
    unit = 2.0 * T * T * hd * B * nh
    print(f'T{T} p{p}: fused fwd {tf:.0f} us ({2 * unit / tf / 1e6:.0f} TF), bwd {tb:.0f} us ({5 * unit / tb / 1e6:.0f} TF algorithmic, 5 products)', flush=True)
