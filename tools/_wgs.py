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
dw = torch.zeros(32, 32, 3, 3, device='cuda'); db = torch.zeros(32, device='cuda')
for (N, H, W) in [(1, 8, 64), (1, 8, 128), (8, 50, 69), (8, 100, 138), (8, 200, 276), (8, 400, 552)]:
    x = torch.randn(N, H, W, 32, device='cuda').to(dt); dy = torch.randn(N, H, W, 32, device='cuda').to(dt); y = torch.empty_like(x)
    w = torch.randn(32, 32, 3, 3, device='cuda') * 0.05; wp = torch.empty(9 * 1024, device='cuda', dtype=dt); lib.conv32_pack_weights(w, wp, 3, 3, 0)
    t1 = timeit(lambda: lib.conv32_wgrad(x, dy, dw, db, N, H, W, 3, 3, 1, 1))
    t2 = timeit(lambda: lib.conv32_fwd(x, wp, db, y, N, H, W, 3, 3, 1, 1))
    t3 = timeit(lambda: lib.act_fwd(x, y, x.numel(), 1, 1))
    print(f'{N}x{H}x{W}: wgrad {t1:.1f} us | fwd {t2:.1f} us | act_fwd {t3:.1f} us')
