"""What a plain device-to-device copy reaches on this box (the practical HBM ceiling the streaming kernels are compared with)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcct_amd._lib import lib, BF16

def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

for mb in (113, 226, 452, 904):
    n = mb * 1000 * 1000 // 2
    x = torch.randn(n, device='cuda').to(torch.bfloat16)
    y = torch.empty_like(x)
    t1 = timeit(lambda: y.copy_(x))
    t2 = timeit(lambda: torch.add(x, x, out=y))
    t3 = timeit(lambda: y.zero_())
    t4 = timeit(lambda: x.float().sum()) if mb <= 226 else float('nan')
    print(f'{mb} MB: copy {t1:.3f} ms = {2 * n * 2 / t1 / 1e6:.0f} GB/s | x+x {t2:.3f} ms = {2 * n * 2 / t2 / 1e6:.0f} GB/s | fill {t3:.3f} ms = {n * 2 / t3 / 1e6:.0f} GB/s')
C = 32
M = 8 * 800 * 1104
x = torch.randn(M, C, device='cuda').to(torch.bfloat16)
y = torch.empty_like(x)
ab = torch.ones(2 * C, device='cuda')
t = timeit(lambda: lib.bn_apply(x, y, M, C, ab, 1, 0, BF16))
print(f'bn_apply 32ch L0: {t:.3f} ms = {2 * x.numel() * 2 / t / 1e6:.0f} GB/s')
