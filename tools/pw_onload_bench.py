"""Timing of the pointwise kernels with an on-load transform (GELU / BatchNorm + Hardswish) against their plain forms, stage-0 shape (1.77 M tokens x 64):
python tools/pw_onload_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tcct_amd._lib import lib
def timeit(fn, iters=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for C, M in ((64, 8 * 400 * 552), (96, 8 * 200 * 276)):
    x = torch.randn(M, C, device='cuda').bfloat16(); res = torch.randn(M, C, device='cuda').bfloat16(); y = torch.empty_like(x)
    dy = torch.randn(M, C, device='cuda').bfloat16(); dx = torch.empty_like(x); yb = torch.randn(M, C, device='cuda').bfloat16()
    w = torch.randn(C, C, device='cuda') / C ** 0.5; b = torch.zeros(C, device='cuda')
    dw = torch.zeros(C, C, device='cuda'); db = torch.zeros(C, device='cuda')
    ab = torch.cat([torch.rand(C) + 0.5, torch.randn(C) * 0.1]).cuda()
    sums = torch.zeros(2 * C, device='cuda', dtype=torch.float64)
    mr = torch.cat([torch.zeros(C), torch.ones(C)]).cuda()
    dg, dbt = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    bsums = torch.ones(2 * C, device='cuda', dtype=torch.float64)
    sp = torch.zeros(2 * C, device='cuda', dtype=torch.float64)
    for _ in range(60): y.copy_(x)
    t = {}
    t['fwd residual (plain)'] = timeit(lambda: lib.pw_fwd_residual(x, w, b, res, None, 1, y, None, M, C, C))
    t['fwd residual + GELU on load'] = timeit(lambda: lib.pw_fwd_gelu_residual(x, w, b, res, None, 1, y, M, C, C))
    t['fwd + stats (plain)'] = timeit(lambda: lib.pw_fwd_bnstats(x, w, None, y, M, C, C, sums, 0))
    t['fwd + stats + BN/hswish on load'] = timeit(lambda: lib.pw_fwd_bnstats_xaff(x, ab, w, None, y, M, C, C, sums))
    t['bwd (plain)'] = timeit(lambda: lib.pw_bwd(x, dy, w, None, dx, dw, db, M, C, C))
    t['bwd + GELU on load'] = timeit(lambda: lib.pw_bwd_gelu(x, dy, w, dx, dw, db, M, C, C))
    red = 2 if C == 64 else -1
    t['bwd BN behind' + (' + reduction' if C == 64 else '')] = timeit(lambda: lib.pw_bwd_bn_sums(x, None, dy, yb, bsums, 0, mr, ab, dg, dbt, 0, w, None, dx, None, dw, None, M, C, C, x if C == 64 else None, ab if C == 64 else None, red, sp if C == 64 else None))
    t['bwd BN behind, x rebuilt on load'] = timeit(lambda: lib.pw_bwd_bn_sums_xaff(x, ab, dy, yb, bsums, 0, mr, ab, dg, dbt, w, None, dx, dw, None, M, C, C, red, sp if C == 64 else None))
    print(f'C={C} M={M}:', '  |  '.join(f'{k} {v:.4f} ms' for k, v in t.items()), flush=True)
