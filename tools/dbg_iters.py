"""does the per-launch time of the 3x3 convolution depend on how long the timing loop runs? (clock ramp / power management)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcct_amd._lib import lib
B, H, W = 8, 800, 1104
x = torch.randn(B, H, W, 32, device='cuda').to(torch.bfloat16)
y = torch.empty_like(x)
w = torch.randn(32, 32, 3, 3, device='cuda') * 0.05
b = torch.zeros(32, device='cuda')
wp = torch.empty(9 * 1024, device='cuda', dtype=torch.bfloat16)
lib.conv32_pack_weights(w, wp, 3, 3, 0)
f = lambda: lib.conv32_fwd(x, wp, b, y, B, H, W, 3, 3, 1, 1)
for iters in (5, 10, 20, 40, 80, 160, 10, 5):
    for _ in range(2):
        f()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        f()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(iters)]
    print(iters, 'mean %.4f' % (sum(ts) / iters), 'first5', ['%.3f' % t for t in ts[:5]], 'last5', ['%.3f' % t for t in ts[-5:]])
    torch.cuda.synchronize()
