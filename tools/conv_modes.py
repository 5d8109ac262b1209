"""A/B of the plain 3x3 32->32 convolution kernels at the bench shape (tcct_conv32_fwd_mode 1 = tiled, 2 = wave-private row streams), interleaved on one box:
python tools/conv_modes.py [scale [stat|-1 [KH KW]]]     (scale 2: the level-1 shape; stat 0 / 1: with the fused BatchNorm statistics of y / LeakyReLU(y))"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tcct_amd._lib import lib
SC = int(sys.argv[1]) if len(sys.argv) > 1 else 1
STAT = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) >= 0 else None
KH, KW = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (3, 3)
B, H, W = 8, 800 // SC, 1104 // SC
x = torch.randn(B, H, W, 32, device='cuda').bfloat16()
y = torch.empty_like(x)
w = torch.randn(32, 32, KH, KW, device='cuda') / 17
b = torch.randn(32, device='cuda')
wp = torch.empty(KH * KW * 1024, device='cuda', dtype=torch.bfloat16)
lib.conv32_pack_weights(w, wp, KH, KW, 0)
sums = torch.zeros(64, device='cuda', dtype=torch.float64)
for _ in range(120):
    y.copy_(x)
def run():
    if STAT is None: lib.conv32_fwd(x, wp, b, y, B, H, W, KH, KW, KH // 2, KW // 2)
    else: lib.conv32_fwd_bnstats(x, wp, b, y, B, H, W, 3, 3, 1, 1, sums, STAT)
def t(mode, iters=30):
    lib.conv32_fwd_mode(mode)
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for rep in range(3):
    print('  '.join(f'mode {m}: {t(m):.4f} ms' for m in (1, 2)), flush=True)
lib.conv32_fwd_mode(0)
