"""Does the 2-D tiled access pattern bound the 3x3 kernels?  Same number of pixels, different image shapes, forward / weight gradient / a plain read:  python tools/wgrad_shapes.py"""
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
w = torch.randn(32, 32, 3, 3, device='cuda') * 0.06
wp = torch.empty(9 * 1024, device='cuda', dtype=torch.bfloat16)
lib.conv32_pack_weights(w, wp, 3, 3, 0)
b = torch.zeros(32, device='cuda')
dw = torch.empty(32, 32, 3, 3, device='cuda'); db = torch.empty(32, device='cuda')
sums = torch.zeros(64, device='cuda', dtype=torch.float64)
mr = torch.zeros(64, device='cuda'); ab = torch.ones(64, device='cuda')
for shp in ((8, 800, 1104), (1, 800, 8832), (1, 6400, 1104), (32, 400, 552), (2, 1600, 2208)):
    N, H, W = shp
    x = torch.randn(N, H, W, 32, device='cuda').bfloat16(); dy = torch.randn(N, H, W, 32, device='cuda').bfloat16(); y = torch.empty_like(x)
    for _ in range(60): y.copy_(x)
    t0 = timeit(lambda: lib.conv32_fwd(x, wp, b, y, N, H, W, 3, 3, 1, 1))
    t1 = timeit(lambda: lib.conv32_wgrad(x, dy, dw, db, N, H, W, 3, 3, 1, 1))
    t2 = timeit(lambda: lib.bn_bwd_reduce(x, dy, N * H * W, 32, mr, ab, 0, 0, sums, 1))
    t3 = timeit(lambda: y.copy_(x))
    print(f'{shp}: conv3x3 fwd {t0:.4f} ms | wgrad {t1:.4f} ms | linear read of both tensors (bn_bwd_reduce) {t2:.4f} ms | copy {t3:.4f} ms', flush=True)
