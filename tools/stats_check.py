"""fused BatchNorm statistics of the 3x3 convolution kernels (row streams / tiled / chain) against fp64 sums of the stored output, bench and test sizes"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcct_amd._lib import lib
torch.manual_seed(0)
for (B, H, W) in [(2, 800, 1104), (8, 800, 1104), (2, 400, 552), (3, 70, 130)]:
    x = (torch.randn((B, H, W, 32), device='cuda') * 0.7 + 0.1).bfloat16()
    w = torch.randn((32, 32, 3, 3), device='cuda') / 17
    b = torch.randn(32, device='cuda') * 0.1
    wp = torch.empty(9 * 1024, device='cuda', dtype=torch.bfloat16)
    lib.conv32_pack_weights(w, wp, 3, 3, 0)
    for mode in (0, 1):
        lib.conv32_fwd_mode(mode)
        for stat in (0, 1):            # activation code in front of the statistics: 0 none, 1 LeakyReLU
            y = torch.empty_like(x)
            sums = torch.zeros(64, device='cuda', dtype=torch.float64)
            lib.conv32_fwd_bnstats(x, wp, b, y, B, H, W, 3, 3, 1, 1, sums, stat)
            torch.cuda.synchronize()
            u = y.double()
            if stat == 1:
                u = torch.where(u > 0, u, 0.01 * u)
            u = u.reshape(-1, 32)
            s1, s2 = u.sum(0), (u * u).sum(0)
            e1 = ((sums[:32] - s1).abs() / s1.abs().clamp_min(1e-9)).max().item()
            e2 = ((sums[32:] - s2).abs() / s2.abs().clamp_min(1e-9)).max().item()
            print((B, H, W), 'fwd_mode', mode, 'stat', stat, 'rel err sum %.2e sumsq %.2e' % (e1, e2), flush=True)
    lib.conv32_fwd_mode(0)
    # chain33 with stats
    y = torch.empty_like(x); mid = torch.empty_like(x)
    sums = torch.zeros(64, device='cuda', dtype=torch.float64)
    lib.conv32_chain33(x, wp, b, mid, wp, b, y, None, B, H, W, sums)
    torch.cuda.synchronize()
    u = y.double(); u = torch.where(u > 0, u, 0.01 * u).reshape(-1, 32)
    print((B, H, W), 'chain33 stats rel err sumsq %.2e' % (((sums[32:] - (u * u).sum(0)).abs() / (u * u).sum(0)).max().item()))
