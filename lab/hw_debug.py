import sys, torch
sys.path.insert(0, '.')
from forwardtacotron_amd import hip as H
torch.manual_seed(0)
rows, C = int(sys.argv[1]), int(sys.argv[2])
x = torch.randn(rows, C, device='cuda'); w1 = torch.randn(C, C, device='cuda') / C ** 0.5; w2 = torch.randn(C, C, device='cuda') / C ** 0.5
b1 = torch.randn(C, device='cuda') * 0.3; b2 = torch.randn(C, device='cuda') * 0.3
out, x12 = H.highway_fwd(x, H.highway_pack(w1, w2), b1, b2, True)
x12r = H.linear_multi_fwd(x, [w1, w2], [b1, b2]); outr = H.highway_gate_fwd(x12r, x)
print('fwd out', float((out - outr).abs().max()), 'x12', float((x12 - x12r).abs().max()))
# a second layer on top: its data gradient with the fused gate gradient of THIS layer
dout2 = torch.randn(rows, C, device='cuda'); x12b = torch.randn(rows, 2 * C, device='cuda'); w3 = torch.randn(C, C, device='cuda') / C ** 0.5; w4 = torch.randn(C, C, device='cuda') / C ** 0.5
d12_top, dx_top = H.highway_gate_bwd(dout2, x12b, out)
# reference: unfused
dxr = dx_top.clone()
H.linear_bwd_data_multi([d12_top.data_ptr(), d12_top.data_ptr() + 4 * C], 2 * C, [w3, w4], dxr, rows, C, accumulate=True)
d12r, dxr2 = H.highway_gate_bwd(dxr, x12r, x)
dxf = dx_top.clone()
d12f = H.highway_bwd_data(d12_top, w3, w4, dxf, below=(x12r, x))
print('bwd d12', float((d12f - d12r).abs().max()), 'dx', float((dxf - dxr2).abs().max()), 'scale', float(d12r.abs().max()))
bad = (d12f - d12r).abs() > 1e-4
print('bad elems', int(bad.sum()), 'rows', bad.any(1).nonzero().flatten()[:10].tolist(), 'cols', bad.any(0).nonzero().flatten()[:20].tolist())
