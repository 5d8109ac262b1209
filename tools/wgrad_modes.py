"""A/B of the 32->32 weight-gradient forms at the bench shape (tcct_conv32_wgrad_mode 0 = rolling rows (3x3) / shifted lines (1 x K, K x 1), 1 = generic),
interleaved on one box:   python tools/wgrad_modes.py [KH KW [scale]]     (scale 2: the level-1 shape)"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tcct_amd._lib import lib
KH, KW = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3, 3)
SC = int(sys.argv[3]) if len(sys.argv) > 3 else 1
B, H, W = 8, 800 // SC, 1104 // SC
x = torch.randn(B, H, W, 32, device='cuda').bfloat16()
dy = torch.randn(B, H, W, 32, device='cuda').bfloat16()
dw = torch.empty(32, 32, KH, KW, device='cuda'); db = torch.empty(32, device='cuda')
for _ in range(120):
    dy.copy_(dy)
def t(mode, iters=30):
    lib.conv32_wgrad_mode(mode)
    for _ in range(5): lib.conv32_wgrad(x, dy, dw, db, B, H, W, KH, KW, KH // 2, KW // 2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): lib.conv32_wgrad(x, dy, dw, db, B, H, W, KH, KW, KH // 2, KW // 2)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
modes = [1, 0, 2] if (KH, KW) == (3, 3) else ([1, 4, 3] if KH * KW in (11, 13) else [1, 0])          # 2: row streams (3x3); 4: shifted lines, 3: one-wave-per-SIMD row streams (13 / 11 taps)
for rep in range(3):
    print('  '.join(f'mode {m}: {t(m):.4f} ms' for m in modes), flush=True)
lib.conv32_wgrad_mode(0)
