import sys, os, torch, time
sys.path.insert(0, '/root/repo')
from tcct_amd._lib import lib
def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e6
dt = torch.bfloat16
lib.set_outputs_prezeroed(1)
for (M, K, N) in [(27600, 160, 160), (27600, 320, 160), (110400, 128, 128), (110400, 256, 160), (27600, 160, 32), (441600, 96, 96), (1766400, 64, 64)]:
    x = torch.randn(M, K, device='cuda').to(dt); w = torch.randn(N, K, device='cuda') * 0.1; b = torch.zeros(N, device='cuda')
    y = torch.empty(M, N, device='cuda', dtype=dt); dy = torch.randn(M, N, device='cuda').to(dt); dx = torch.empty_like(x)
    dw = torch.zeros(N, K, device='cuda'); db = torch.zeros(N, device='cuda')
    t1 = timeit(lambda: lib.pw_fwd(x, w, b, y, M, K, N, 0, 1))
    t2 = timeit(lambda: lib.pw_fwd(dy, w, None, dx, M, N, K, 1, 1))
    t3 = timeit(lambda: lib.pw_wgrad(x, dy, dw, db, M, K, N)) if N <= 160 and N % 32 == 0 else 0
    print(f'M={M} K={K} N={N}: fwd {t1:.1f} us | dgrad {t2:.1f} us | wgrad {t3:.1f} us')
