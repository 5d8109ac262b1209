"""tcct_bilinear_bwd of the level-0 head's gradient (fp32 [8,800,1104,5] -> [8,400,552,5], align_corners=False) and of the decoder's (bf16 32 channels), HIP events.

    python tools/bilinear_bwd_bench.py   (TCCT_LIB_PATH=ab/libtcct_REV.so for the other arm)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcct_amd._lib import lib
from tools.kbench import timeit

dev = 'cuda'
spin = torch.empty(8, 800, 1104, 32, device=dev, dtype=torch.bfloat16)
for _ in range(120):
    spin.copy_(spin)
torch.manual_seed(0)
for C, dt, code in ((5, torch.float32, 0), (32, torch.bfloat16, 1)):
    dy = torch.randn(8, 800, 1104, C, device=dev).to(dt)
    dx = torch.empty(8, 400, 552, C, device=dev, dtype=dt)
    mb = (dy.numel() + dx.numel()) * dy.element_size() / 1e6
    ref = None
    for form in (0, 1):
        lib.bilinear_bwd_x2(form)
        ms = timeit(lambda: lib.bilinear_bwd(dy, dx, 8, 400, 552, C, 800, 1104, 0, code), iters=30, warm=3)
        d = dx.float().clone()
        err = 0.0 if ref is None else float((d - ref).abs().max() / ref.abs().max())
        ref = d if ref is None else ref
        print(f'bilinear_bwd x2, {C} channels {dt}, {"separable lane-exchange kernel" if form else "tiled gather kernel"}: {ms * 1e3:7.1f} us, {mb / ms / 1e3:.2f} TB/s on dy + dx '
              f'({mb:.0f} MB); max |difference| / max |dx| against the gather kernel {err:.2e}')
    lib.bilinear_bwd_x2(1)
