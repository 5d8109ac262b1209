import sys, os, torch, time
sys.path.insert(0, '/root/repo')
from tcct_amd._lib import lib
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3
dt = torch.bfloat16
for (M, C) in [(1766400, 64), (7065600, 32), (441600, 96)]:
    x = torch.randn(M, C, device='cuda').to(dt); dy = torch.randn(M, C, device='cuda').to(dt); y = torch.empty_like(x)
    ab = torch.ones(2 * C, device='cuda'); mr = torch.zeros(2 * C, device='cuda'); mr[C:] = 1
    sums = torch.zeros(2 * C, device='cuda', dtype=torch.float64); dg = torch.empty(C, device='cuda'); db = torch.empty(C, device='cuda')
    gb = x.numel() * 2 / 1e9
    for (pre, post) in [(0, 0), (1, 0), (0, 2), (0, 1)]:
        t1 = timeit(lambda: lib.bn_apply(x, y, M, C, ab, pre, post, 1))
        t2 = timeit(lambda: lib.bn_bwd_reduce(x, dy, M, C, mr, ab, pre, post, sums, 1))
        t3 = timeit(lambda: lib.bn_bwd_apply(x, dy, y, M, C, mr, ab, ab, sums, pre, post, dg, db, 1))
        t4 = timeit(lambda: lib.bn_stats(x, M, C, pre, sums, 1))
        print(f'M={M} C={C} pre={pre} post={post}: apply {t1:.3f} ms {2*gb/t1*1e3:.0f} GB/s | bwd_reduce {t2:.3f} {2*gb/t2*1e3:.0f} | bwd_apply {t3:.3f} {3*gb/t3*1e3:.0f} | stats {t4:.3f} {gb/t4*1e3:.0f}')
