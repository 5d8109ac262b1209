import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tcct_amd._lib import lib
torch.manual_seed(0)
N, H, W, stride, post = 1, 4, 32, 1, 0        # M = 128: exactly one tile
x = torch.zeros(N, 3, H, W)
x[0, 0] = torch.arange(H * W).view(H, W).float()        # channel 0 = pixel index (exact in bf16 up to 256)
w = torch.zeros(32, 3, 3, 3); w[:, 0, 1, 1] = 1.0       # y[p][c] = x0[p]
b = torch.zeros(32)
x4 = F.pad(x, (0, 0, 0, 0, 0, 1)).permute(0, 2, 3, 1).contiguous().bfloat16().cuda()
ab = torch.cat([torch.ones(32), torch.zeros(32)]).cuda()
got = torch.zeros(128)
for p in range(128):
    dz = torch.zeros(1, H, W, 32); dz.view(-1, 32)[p, p % 32] = 1.0
    raw = torch.zeros(64, dtype=torch.float64, device='cuda')
    lib.c3_bn_bwd_reduce(x4, w.cuda(), b.cuda(), dz.bfloat16().cuda(), N, H, W, stride, ab, raw, post)
    got[p] = raw[32 + p % 32].item()
print('y seen at pixel p (should be p):')
print(got.view(8, 16).int())
