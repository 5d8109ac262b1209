"""debug: stage-by-stage check of csrc/c3_bn.hip against torch (run on the GPU box)"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tcct_amd._lib import lib
torch.manual_seed(0)
N, H, W, stride, post = 2, 18, 26, 1, 0
x = torch.randn(N, 3, H, W).bfloat16().float()
w = (torch.randn(32, 3, 3, 3) / 27 ** 0.5).bfloat16().float()
b = torch.randn(32)
g, be = torch.rand(32) + 0.5, torch.randn(32)
y = F.conv2d(x, w, b, stride, 1)
M = y.numel() // 32
mean, var = y.mean((0, 2, 3)), y.var((0, 2, 3), unbiased=False)
rstd = 1 / torch.sqrt(var + 1e-5)
a, bb = g * rstd, be - mean * g * rstd
gz = torch.randn_like(y).bfloat16().float()
x4 = F.pad(x, (0, 0, 0, 0, 0, 1)).permute(0, 2, 3, 1).contiguous().bfloat16().cuda()
dz = gz.permute(0, 2, 3, 1).contiguous().bfloat16().cuda()
ab = torch.cat([a, bb]).cuda()
mr = torch.cat([mean, rstd]).cuda()
raw = torch.zeros(64, dtype=torch.float64, device='cuda')
lib.c3_bn_bwd_reduce(x4, w.cuda(), b.cuda(), dz, N, H, W, stride, ab, raw, post)
S1 = gz.sum((0, 2, 3)).double()
S2 = (gz * y).sum((0, 2, 3)).double()
print('raw S1 err', (raw[:32].cpu() - S1).abs().max().item(), 'of', S1.abs().max().item())
print('raw S2 err', (raw[32:].cpu() - S2).abs().max().item(), 'of', S2.abs().max().item())
coef = torch.empty(160, device='cuda')
dg, db = torch.empty(32, device='cuda'), torch.empty(32, device='cuda')
lib.bn_bwd_coef(raw, 1, M, 32, mr, ab, coef, dg, db)
xh = (y - mean.view(1, -1, 1, 1)) * rstd.view(1, -1, 1, 1)
dy = a.view(1, -1, 1, 1) * (gz - gz.mean((0, 2, 3), keepdim=True) - xh * (gz * xh).mean((0, 2, 3), keepdim=True))
c = coef.cpu()
dy2 = c[:32].view(1, -1, 1, 1) * gz + c[32:64].view(1, -1, 1, 1) * y + c[64:96].view(1, -1, 1, 1)
print('dy from coef err', (dy2 - dy).abs().max().item(), 'of', dy.abs().max().item())
wr = w.clone().requires_grad_(True)
F.conv2d(x, wr, None, stride, 1).backward(dy)
for name, dyt in (('zero-c2c3 (c1=1)', None),):
    pass
dw = torch.zeros(32, 3, 3, 3, device='cuda'); dbias = torch.zeros(32, device='cuda')
lib.c3_bn_bwd_wgrad(x4, w.cuda(), b.cuda(), dz, N, H, W, stride, coef, dw, dbias, post)
e = (dw.cpu() - wr.grad)
print('dw err max', e.abs().max().item(), 'of', wr.grad.abs().max().item(), 'rel-L2', (e.norm() / wr.grad.norm()).item())
print('per-co rel err', [round((e[i].norm() / wr.grad[i].norm()).item(), 3) for i in range(32)])
# identity coefficients: dy = dz -> plain weight gradient
coef1 = coef.clone(); coef1[:32] = 1; coef1[32:96] = 0
dw1 = torch.zeros(32, 3, 3, 3, device='cuda')
lib.c3_bn_bwd_wgrad(x4, w.cuda(), b.cuda(), dz, N, H, W, stride, coef1, dw1, dbias, post)
wr2 = w.clone().requires_grad_(True)
F.conv2d(x, wr2, None, stride, 1).backward(gz)
print('dy=dz: dw rel-L2', ((dw1.cpu() - wr2.grad).norm() / wr2.grad.norm()).item())
# dy = y: c1 = 0, c2 = 1
coef2 = coef.clone(); coef2[:32] = 0; coef2[32:64] = 1; coef2[64:96] = 0
dw2 = torch.zeros(32, 3, 3, 3, device='cuda')
lib.c3_bn_bwd_wgrad(x4, w.cuda(), b.cuda(), dz, N, H, W, stride, coef2, dw2, dbias, post)
wr3 = w.clone().requires_grad_(True)
F.conv2d(x, wr3, None, stride, 1).backward(y.detach())
e3 = dw2.cpu() - wr3.grad
print('dy=y: dw rel-L2', (e3.norm() / wr3.grad.norm()).item(), 'per-co', [round((e3[i].norm() / wr3.grad[i].norm()).item(), 3) for i in range(0, 32, 4)])
