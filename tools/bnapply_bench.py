"""k_bn_apply at the ViT shapes of the step: plain / Hardswish / + residual (python tools/bnapply_bench.py)"""
import sys, torch
sys.path.insert(0, '.')
from tcct_amd._lib import lib
def t(fn, iters=30):
    for _ in range(5): fn()
    e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters*1e3
for (M,C) in [(1766400,64),(1766400,96),(441600,96),(441600,128)]:
    x = torch.randn(M, C, device='cuda').bfloat16(); r = torch.randn(M, C, device='cuda').bfloat16(); y = torch.empty_like(x)
    sums = torch.zeros(2*C, device='cuda', dtype=torch.float64); sums[C:] = M
    g, b = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
    mr, ab = torch.empty(2*C, device='cuda'), torch.empty(2*C, device='cuda')
    out=[]
    for nm, res, post in (('plain', None, 0), ('hswish', None, 2), ('hswish+res', r, 2), ('res', r, 0)):
        us = t(lambda: lib.bn_apply_train(x, res, y, M, C, sums, g, b, 1e-5, 0.1, None, None, None, mr, ab, 0, post, 1))
        nb = (3 if res is not None else 2) * M * C * 2
        out.append(f'{nm} {us:.1f} us ({nb/us/1e6:.2f} TB/s)')
    print(f'({M},{C}): ' + '  '.join(out))
