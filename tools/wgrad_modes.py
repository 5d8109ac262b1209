"""A/B of the two 3x3 32->32 weight-gradient forms at the bench shape (tcct_conv32_wgrad_mode 0 = rolling rows, 1 = generic), interleaved on one box:
python tools/wgrad_modes.py"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tcct_amd._lib import lib
B, H, W = 8, 800, 1104
x = torch.randn(B, H, W, 32, device='cuda').bfloat16()
dy = torch.randn(B, H, W, 32, device='cuda').bfloat16()
dw = torch.empty(32, 32, 3, 3, device='cuda'); db = torch.empty(32, device='cuda')
for _ in range(120):
    dy.copy_(dy)
def t(mode, iters=30):
    lib.conv32_wgrad_mode(mode)
    for _ in range(5): lib.conv32_wgrad(x, dy, dw, db, B, H, W, 3, 3, 1, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): lib.conv32_wgrad(x, dy, dw, db, B, H, W, 3, 3, 1, 1)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
modes = [int(m) for m in sys.argv[1:]] or [1, 0]
for rep in range(3):
    print('  '.join(f'mode {m}: {t(m):.4f} ms' for m in modes), flush=True)
lib.conv32_wgrad_mode(0)
