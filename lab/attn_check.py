"""fused attention (bf16) against float64 attention on the bf16-rounded operands, and against the unfused bf16 path"""
import math, os, sys, torch
sys.path.insert(0, '.')
from forwardtacotron_amd import hip as H
torch.manual_seed(0)
def run(B, T, d, nh, p_drop, ragged=True):
    hd = d // nh
    qkv = torch.randn(B, T, 3 * d, device='cuda') * 0.7
    lens = torch.randint(max(1, T // 2), T + 1, (B,)); lens[0] = T
    key_pad = (torch.arange(T)[None, :] >= lens[:, None]).to(torch.uint8).cuda() if ragged else None
    scale = 1.0 / math.sqrt(hd)
    seed = 1234567
    att, lse2 = H.attn_fwd(qkv, key_pad, nh, scale, p_drop, seed)
    datt = torch.randn(B, T, d, device='cuda')
    dqkv = H.attn_bwd(qkv, att, datt, key_pad, lse2, nh, scale, p_drop, seed)
    # float64 reference on bf16-rounded operands
    q, k, v = [t.bfloat16().double().reshape(B, T, nh, hd).permute(0, 2, 1, 3) for t in qkv.split(d, dim=-1)]
    q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
    s = (q @ k.transpose(-1, -2)) * scale
    if key_pad is not None:
        s = s.masked_fill(key_pad.bool()[:, None, None, :], float('-inf'))
    P = torch.softmax(s, dim=-1)
    if p_drop > 0:
        # the library's counter-based mask, through ft_dropout on an index-shaped tensor: element i kept iff hash(seed, i) >= p
        ones = torch.ones(B * nh * T * T, device='cuda')
        keep = (H.dropout(ones, p_drop, seed) > 0).double().reshape(B, nh, T, T)
        Pd = P * keep / (1 - p_drop)
    else:
        Pd = P
    o = (Pd @ v).permute(0, 2, 1, 3).reshape(B, T, d)
    (o * datt.double()).sum().backward()
    ref_dqkv = torch.cat([t.grad.permute(0, 2, 1, 3).reshape(B, T, d) for t in (q, k, v)], dim=-1)
    e_fwd = float((att.double() - o.detach()).abs().max()) / float(o.abs().max())
    e_bwd = [float((a.double() - r).abs().max()) / float(r.abs().max()) for a, r in zip(dqkv.split(d, dim=-1), ref_dqkv.split(d, dim=-1))]
    print(f'B{B} T{T} d{d} nh{nh} p{p_drop}: fwd rel err {e_fwd:.2e}  dq/dk/dv rel err {[f"{e:.2e}" for e in e_bwd]}', flush=True)
for cfg in [(2, 70, 128, 2, 0.0), (2, 70, 128, 2, 0.1), (3, 200, 256, 2, 0.0), (2, 333, 256, 2, 0.1), (1, 64, 256, 2, 0.0), (2, 129, 128, 2, 0.0)]:
    run(*cfg)
run(2, 100, 256, 2, 0.0, ragged=False)
